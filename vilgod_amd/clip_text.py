"""Text side of `ClipWrapper.__init__`: the 24 prompts -> L2-normalised text features, computed ONCE at
start-up with plain torch (not a kernel, not on the hot path) and cached next to the checkpoint.

Mirrors
  clip.tokenize                     third_party/CLIP/clip/clip.py:197-237        (sot + BPE + eot, zero padded to 77)
  SimpleTokenizer                   third_party/CLIP/clip/simple_tokenizer.py:62-132 (byte-level BPE over the shipped merges file)
  CLIP.encode_text                  third_party/CLIP/clip/model.py:343-356       (token + positional embedding, causal
                                    transformer of ResidualAttentionBlocks :171-192, ln_final, row of the eot token @ text_projection)
  ClipWrapper.__init__ text side    src/utils/clip_utils.py:22-26                (prompt_template.format(cls), normalise)

The merges file `bpe_simple_vocab_16e6.txt.gz` is data of the CLIP package the reference vendors; it is looked up next to an
installed `clip` package (the reference's PYTHONPATH contract, README.md:130-133) or taken from `bpe_path`.
"""
import gzip
import html
import importlib.util
import os

import numpy as np
import torch


def find_bpe_vocab(bpe_path=None):
    if bpe_path and os.path.exists(bpe_path):
        return bpe_path
    spec = importlib.util.find_spec('clip')                     # locates the package without importing it
    if spec is not None and spec.submodule_search_locations:
        p = os.path.join(list(spec.submodule_search_locations)[0], 'bpe_simple_vocab_16e6.txt.gz')
        if os.path.exists(p):
            return p
    return None


def _byte_alphabet():
    """The reversible byte <-> printable-unicode table of the GPT-2 / CLIP BPE."""
    keep = list(range(ord('!'), ord('~') + 1)) + list(range(ord('¡'), ord('¬') + 1)) + list(range(ord('®'), ord('ÿ') + 1))
    table, extra = {}, 0
    for b in keep:
        table[b] = chr(b)
    for b in range(256):
        if b not in table:
            table[b] = chr(256 + extra)
            extra += 1
    order = keep + [b for b in range(256) if b not in keep]
    return table, [table[b] for b in order]


class BpeTokenizer:
    def __init__(self, bpe_path):
        import regex
        self.byte_to_char, alphabet = _byte_alphabet()
        with gzip.open(bpe_path) as f:
            lines = f.read().decode('utf-8').split('\n')
        merges = [tuple(l.split()) for l in lines[1:49152 - 256 - 2 + 1]]
        vocab = alphabet + [c + '</w>' for c in alphabet] + [''.join(m) for m in merges] + ['<|startoftext|>', '<|endoftext|>']
        self.token_id = {t: i for i, t in enumerate(vocab)}
        self.rank = {m: i for i, m in enumerate(merges)}
        self.sot, self.eot = self.token_id['<|startoftext|>'], self.token_id['<|endoftext|>']
        self.word_re = regex.compile(r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""",
                                     regex.IGNORECASE)
        self._ws_re = regex.compile(r'\s+')
        self._cache = {}

    def _merge(self, word):
        """Greedy BPE: repeatedly join the adjacent pair with the lowest merge rank."""
        if word in self._cache:
            return self._cache[word]
        parts = list(word[:-1]) + [word[-1] + '</w>']
        while len(parts) > 1:
            ranked = [(self.rank.get((a, b), None), i) for i, (a, b) in enumerate(zip(parts[:-1], parts[1:]))]
            ranked = [(r, i) for r, i in ranked if r is not None]
            if not ranked:
                break
            best = min(ranked)[0]
            first, second = next((a, b) for a, b in zip(parts[:-1], parts[1:]) if self.rank.get((a, b)) == best)
            out, i = [], 0
            while i < len(parts):
                if i < len(parts) - 1 and parts[i] == first and parts[i + 1] == second:
                    out.append(first + second)
                    i += 2
                else:
                    out.append(parts[i])
                    i += 1
            parts = out
        self._cache[word] = parts
        return parts

    def encode(self, text):
        if not text.isascii():
            try:
                import ftfy
                text = ftfy.fix_text(text)
            except ImportError as e:
                raise RuntimeError('non-ASCII prompt: the reference cleans it with ftfy, which is not installed') from e
        text = html.unescape(html.unescape(text)).strip()
        text = self._ws_re.sub(' ', text).strip().lower()
        ids = []
        for w in self.word_re.findall(text):
            chars = ''.join(self.byte_to_char[b] for b in w.encode('utf-8'))
            ids.extend(self.token_id[p] for p in self._merge(chars))
        return ids


def tokenize(tokenizer, texts, context_length=77):
    """-> int64 [n, context_length]: sot, BPE ids, eot, zero padding (clip.py:197-237, truncate=False)."""
    out = np.zeros((len(texts), context_length), np.int64)
    for i, t in enumerate(texts):
        ids = [tokenizer.sot] + tokenizer.encode(t) + [tokenizer.eot]
        if len(ids) > context_length:
            raise RuntimeError(f'Input {t} is too long for context length {context_length}')
        out[i, :len(ids)] = ids
    return out


def load_text_state_dict(path):
    """The text-tower tensors of a CLIP checkpoint (TorchScript archive or plain state dict), float32."""
    try:
        sd = torch.jit.load(path, map_location='cpu').state_dict()
    except RuntimeError:
        sd = torch.load(path, map_location='cpu')
    keep = ('token_embedding.', 'positional_embedding', 'transformer.', 'ln_final.', 'text_projection')
    return {k: v.float() for k, v in sd.items() if k.startswith(keep)}


@torch.no_grad()
def encode_text(sd, tokens, device='cpu'):
    """model.py:343-356 on a state dict.  tokens int64 [n, ctx] -> float32 [n, embed] (NOT normalised)."""
    sd = {k: v.to(device=device, dtype=torch.float32) for k, v in sd.items()}
    tok = torch.as_tensor(np.asarray(tokens), dtype=torch.long, device=device)
    x = sd['token_embedding.weight'][tok] + sd['positional_embedding'][:tok.shape[1]]
    n, L, W = x.shape
    heads = max(W // 64, 1)
    dh = W // heads
    mask = torch.full((L, L), float('-inf'), device=device).triu_(1)                # model.py:320-326 build_attention_mask
    ln = lambda t, p: torch.nn.functional.layer_norm(t, (W,), sd[p + '.weight'], sd[p + '.bias'], 1e-5)
    layers = len([k for k in sd if k.endswith('attn.in_proj_weight')])
    for i in range(layers):
        p = f'transformer.resblocks.{i}.'
        h = ln(x, p + 'ln_1')
        qkv = h @ sd[p + 'attn.in_proj_weight'].t() + sd[p + 'attn.in_proj_bias']
        q, k, v = (t.reshape(n, L, heads, dh).transpose(1, 2) for t in qkv.chunk(3, dim=-1))
        a = torch.softmax((q * dh ** -0.5) @ k.transpose(-1, -2) + mask, dim=-1) @ v
        a = a.transpose(1, 2).reshape(n, L, W)
        x = x + a @ sd[p + 'attn.out_proj.weight'].t() + sd[p + 'attn.out_proj.bias']
        h = ln(x, p + 'ln_2')
        h = h @ sd[p + 'mlp.c_fc.weight'].t() + sd[p + 'mlp.c_fc.bias']
        h = h * torch.sigmoid(1.702 * h)                                              # QuickGELU, model.py:166-168
        x = x + h @ sd[p + 'mlp.c_proj.weight'].t() + sd[p + 'mlp.c_proj.bias']
    x = ln(x, 'ln_final')
    eot = tok.argmax(dim=-1)                                                          # the eot token has the highest id
    return (x[torch.arange(n, device=device), eot] @ sd['text_projection']).float().cpu()


def text_features(ckpt_path, prompts, bpe_path=None, device='cpu'):
    """clip_utils.py:22-26: normalised features of the prompts (float32 [n, embed]; the reference computes them in the
    model's dtype -- fp16 on a GPU -- so its values differ from these in the 3rd-4th digit)."""
    vocab = find_bpe_vocab(bpe_path)
    if vocab is None:
        raise FileNotFoundError('bpe_simple_vocab_16e6.txt.gz not found: put third_party/CLIP on PYTHONPATH (README.md:130-133 '
                                'of the reference) or pass its path (paths.clip_bpe)')
    tok = tokenize(BpeTokenizer(vocab), list(prompts))
    f = encode_text(load_text_state_dict(ckpt_path), tok, device=device)
    return (f / f.norm(dim=-1, keepdim=True)).numpy()
