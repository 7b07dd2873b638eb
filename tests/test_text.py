"""Text side of ClipWrapper: prompts -> token ids -> CLIP.encode_text -> normalised features.
Golden vectors: tests/golden/text_golden.npz, produced by the REFERENCE's tokenizer and model code on a small seeded text
tower (tests/golden/make_golden.py text).  The tokenizer needs the CLIP package's merges file (data of that package, not
vendored here): its test runs wherever the file can be found (next to an importable `clip`, $CLIP_BPE, or the build container's
reference checkout) and is skipped elsewhere; the transformer test needs nothing."""
import os

import numpy as np
import pytest

from vilgod_amd import clip_text, clip_weights as cw


@pytest.fixture(scope='module')
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, 'text_golden.npz'))


def test_text_tower_matches_reference(gold):
    sd = cw.synthetic_text_weights(0, width=int(gold['width']), layers=int(gold['layers']), embed=int(gold['embed']))
    f = clip_text.encode_text(sd, gold['tokens']).numpy()
    assert f.shape == gold['features'].shape
    assert np.abs(f - gold['features']).max() < 2e-5


def _bpe():
    for p in (os.environ.get('CLIP_BPE'), clip_text.find_bpe_vocab(),
              '/root/reference/third_party/CLIP/clip/bpe_simple_vocab_16e6.txt.gz'):
        if p and os.path.exists(p):
            return p
    return None


@pytest.mark.skipif(_bpe() is None, reason='bpe_simple_vocab_16e6.txt.gz (CLIP package data) not available')
def test_tokenizer_matches_reference(gold):
    tk = clip_text.BpeTokenizer(_bpe())
    tok = clip_text.tokenize(tk, [str(t) for t in gold['texts']])
    assert np.array_equal(tok, gold['tokens'])
    assert tok[0, 0] == 49406 and tok[0].max() == 49407            # <|startoftext|> ... <|endoftext|>
    with pytest.raises(RuntimeError):
        clip_text.tokenize(tk, ['word ' * 100])


@pytest.mark.skipif(_bpe() is None, reason='bpe_simple_vocab_16e6.txt.gz (CLIP package data) not available')
def test_clipwrapper_computes_and_caches_text_features(tmp_path):
    """Real-checkpoint mode end to end on the host side: a (small) state-dict checkpoint without a cached feature file ->
    features computed with torch from the text tower, normalised, cached next to the checkpoint."""
    import torch
    sd = {**{'visual.' + k: v for k, v in cw.synthetic_vit_weights(0, width=64, layers=1, heads=1, patch=16, resolution=32, output_dim=32).items()},
          **cw.synthetic_text_weights(1, width=64, layers=1, embed=32)}
    ckpt = tmp_path / 'ViT-tiny.pt'
    torch.save(sd, ckpt)
    prompts = ['a point representation of a car', 'a point representation of a person']
    f = clip_text.text_features(str(ckpt), prompts, bpe_path=_bpe())
    assert f.shape == (2, 32) and np.allclose(np.linalg.norm(f, axis=1), 1.0, atol=1e-6)
    tk = clip_text.BpeTokenizer(_bpe())
    want = clip_text.encode_text(cw.synthetic_text_weights(1, width=64, layers=1, embed=32), clip_text.tokenize(tk, prompts))
    want = want / want.norm(dim=-1, keepdim=True)
    assert np.allclose(f, want.numpy(), atol=1e-6)
