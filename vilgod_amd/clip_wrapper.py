"""Host side of CLIP zero-shot classification (mirror of the reference's
`src/utils/clip_utils.py::ClipWrapper`: same constructor arguments, same `predict_clip_labels`
return values) on top of the HIP ViT of csrc/vit.hip.

Differences from the reference, all on purpose:
  * crops arrive as a CUDA tensor straight from the renderer (no PIL round trip; the uint8
    quantisation and CLIP normalisation already happened in the render kernel);
  * all crops of a frame are encoded in one call (`split_size` only bounds the workspace);
  * without a checkpoint on disk (`<model_path>/<model_name>`) seeded synthetic weights and text
    features are used (bench / tests) -- stated loudly in `self.weights_source`.
"""
import ctypes
import os

import numpy as np
import torch

from . import clip_weights
from ._lib import lib, ptr, stream_ptr, check

DTYPES = {'f32': 0, 'f16': 1}


class VitEncoder:
    """Owns a vg_vit handle + workspace."""

    def __init__(self, weights, dtype='f16', device='cuda'):
        cfg = clip_weights.infer_config(weights)
        self.cfg = cfg
        self.dtype = dtype
        self.device = torch.device(device)
        h = ctypes.c_void_p()
        check(lib.vg_vit_create(ctypes.byref(h), cfg['width'], cfg['layers'], cfg['heads'], cfg['patch'],
                                cfg['resolution'], cfg['output_dim'], DTYPES[dtype]), 'vg_vit_create')
        self._h = h
        # the per-channel normalisation the single-channel patch rows fold into the patch embedding: the renderer's own constants
        from .projection import CLIP_MEAN, CLIP_STD
        check(lib.vg_vit_set_input_norm(h, (ctypes.c_float * 3)(*CLIP_MEAN), (ctypes.c_float * 3)(*CLIP_STD)), 'vg_vit_set_input_norm')
        with torch.cuda.device(self.device):
            for name, t in weights.items():
                t = t.detach().to(torch.float32).contiguous().cpu()
                check(lib.vg_vit_set_weight(self._h, name.encode(), ctypes.c_void_p(t.data_ptr()), t.numel()),
                      f'vg_vit_set_weight({name})')
        self._ws = None
        self._ws_crops = 0
        self._owner = None               # set on views: keeps the owning encoder (and its handle) alive

    def view(self):
        """A second encoder object on the SAME weights (the device weights are read-only during encode) with its own
        workspace: what a worker stream needs.  The owner must not enable profiling while views encode concurrently."""
        import copy
        v = copy.copy(self)
        v._ws, v._ws_crops, v._owner = None, 0, self
        return v

    def __del__(self):
        h = getattr(self, '_h', None)
        if h is not None and lib is not None and getattr(self, '_owner', None) is None:
            lib.vg_vit_destroy(h)
        self._h = None

    def _workspace(self, n):
        if self._ws is None or n > self._ws_crops:
            # sized for the next multiple of 64 crops: a stream of frames meets a slightly larger crop count every few frames, and
            # every exact-size regrowth was a fresh ~1.5 GB allocation + fill on the worker's stream (and the old block stays in the
            # caching allocator's pool)
            # (round 5: + an eighth of headroom -- a worker whose first frames held 310 crops got a 320-crop workspace and paid the
            # regrowth, a device-synchronising 1.5 GB hipMalloc, on its first 330-crop frame; 288 GB of HBM hold the headroom of every worker)
            cap = (n + max(32, n // 8) + 63) // 64 * 64
            nbytes = lib.vg_vit_workspace_bytes(self._h, cap)
            self._ws = None
            self._ws = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
            self._ws_crops = cap
        return self._ws

    def profile(self, on=True):
        check(lib.vg_vit_profile(self._h, 1 if on else 0), 'vg_vit_profile')

    def profile_read(self, kind=-1):
        """-> (launches, total ms, total algorithmic FLOPs) of the projection GEMMs since profile(True);
        kind 1: k_gemm_f16_pp64 launches only, 0: the fallback kernels only, -1: all."""
        n, ms, fl = ctypes.c_int32(0), ctypes.c_double(0), ctypes.c_double(0)
        check(lib.vg_vit_profile_read_kind(self._h, int(kind), ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl)),
              'vg_vit_profile_read_kind')
        return n.value, ms.value, fl.value

    def encode(self, crops, stream=None):
        """crops: [n,3,res,res] float32 or float16 CUDA tensor, or patch rows (see `encode_patches`)
        -> [n,output_dim] float32 features."""
        assert crops.is_cuda and crops.is_contiguous()
        n = crops.shape[0]
        feat = torch.empty((n, self.cfg['output_dim']), dtype=torch.float32, device=crops.device)
        if n == 0:
            return feat
        kind = {torch.float32: 0, torch.float16: 1}[crops.dtype]
        ws = self._workspace(n)
        check(lib.vg_vit_encode(self._h, ptr(crops), kind, n, ptr(ws), ptr(feat), stream_ptr(stream)), 'vg_vit_encode')
        return feat


def _encode_patches(self, patches, n, stream=None):
    """patches: f16 [rows>=n*196 (multiple of 256), 768] from RealisticProjection.render_frame(out='patch16'), or [rows, 256]
    single-channel rows from out='patch16c1' (the tower folds the three identical channels and their normalisation, input_kind 3)."""
    assert patches.is_cuda and patches.dtype == torch.float16 and self.dtype == 'f16' and patches.shape[1] in (768, 256)
    feat = torch.empty((n, self.cfg['output_dim']), dtype=torch.float32, device=patches.device)
    if n:
        ws = self._workspace(n)
        kind = 2 if patches.shape[1] == 768 else 3
        check(lib.vg_vit_encode(self._h, ptr(patches), kind, n, ptr(ws), ptr(feat), stream_ptr(stream)), 'vg_vit_encode')
    return feat


VitEncoder.encode_patches = _encode_patches


class GraphClassifier:
    """One worker's captured classification (include/vilgod_hip.h vg_vit_classify_graph): persistent patch / workspace / feature /
    score buffers and a cache of one hipGraph per distinct crop count.  The renderer writes the frame's patch rows into
    `patch_buffer(n)`; `classify(n)` replays (or, the first time a crop count shows up, captures) the graph of the ViT encode +
    scores on the current stream and returns views of the persistent outputs -- valid until the worker's next frame."""

    BUCKET = 8          # crop counts are rounded up to a multiple of this: a real stream has a new crop count almost every frame, a
                        # graph per exact count would be captured (~150 nodes + instantiate) for most frames.  The rows of the
                        # padding crops hold finite data of an earlier frame; their outputs are never read (<= 7 crops of ~330: ~1 %)

    def __init__(self, encoder, text_features, max_crops=512, max_graphs=32, patch_width=256):
        assert encoder.dtype == 'f16' and encoder.cfg['patch'] == 16 and encoder.cfg['resolution'] == 224 and patch_width in (256, 768)
        self.enc, self.text = encoder, text_features
        self.patch_width = int(patch_width)          # 256: single-channel rows (render 'patch16c1', input_kind 3); 768: 'patch16
        self.device = encoder.device
        self.max_graphs = int(max_graphs)
        self._cache = ctypes.c_void_p()
        check(lib.vg_graph_cache_create(ctypes.byref(self._cache)), 'vg_graph_cache_create')
        check(lib.vg_graph_cache_limit(self._cache, self.max_graphs), 'vg_graph_cache_limit')
        self.cap = 0
        self._alloc(max_crops)

    def _alloc(self, n):
        """(Re)allocate every buffer a graph node points to; graphs of the old buffers are dropped."""
        if self.cap:
            lib.vg_graph_cache_destroy(self._cache)
            self._cache = ctypes.c_void_p()
            check(lib.vg_graph_cache_create(ctypes.byref(self._cache)), 'vg_graph_cache_create')
            check(lib.vg_graph_cache_limit(self._cache, self.max_graphs), 'vg_graph_cache_limit')
        self.cap = (int(n) + self.BUCKET - 1) // self.BUCKET * self.BUCKET
        rows = (self.cap * 196 + 255) // 256 * 256
        K = self.text.shape[0]
        self.patches = torch.zeros((rows, self.patch_width), dtype=torch.float16, device=self.device)
        self.ws = torch.zeros(int(lib.vg_vit_workspace_bytes(self.enc._h, self.cap)), dtype=torch.uint8, device=self.device)
        self.feat = torch.empty((self.cap, self.enc.cfg['output_dim']), dtype=torch.float32, device=self.device)
        self.probs = torch.empty((self.cap, K), dtype=torch.float32, device=self.device)
        self.top1 = torch.empty((self.cap,), dtype=torch.int32, device=self.device)
        self.score = torch.empty((self.cap,), dtype=torch.float32, device=self.device)

    def __del__(self):
        c = getattr(self, '_cache', None)
        if c is not None and lib is not None:
            lib.vg_graph_cache_destroy(c)
            self._cache = None

    def patch_buffer(self, n):
        if n > self.cap:
            torch.cuda.current_stream(self.device).synchronize()
            self._alloc(max(n, int(self.cap * 1.5)))
        return self.patches

    def classify(self, n, text_features=None):
        """-> (probs [n,K], top1 [n], score [n]) views of the persistent outputs."""
        if text_features is not None and text_features is not self.text:
            assert text_features.shape == self.text.shape
            self.text = text_features                # (its pointer is part of the graph key: new features -> new graphs)
        nb = min(self.cap, (n + self.BUCKET - 1) // self.BUCKET * self.BUCKET)
        check(lib.vg_vit_classify_graph(self.enc._h, self._cache, ptr(self.patches), 2 if self.patch_width == 768 else 3, nb, ptr(self.ws), ptr(self.feat), ptr(self.text),
                                        self.feat.shape[1], self.text.shape[0], ptr(self.probs), ptr(self.top1), ptr(self.score),
                                        stream_ptr()), 'vg_vit_classify_graph')
        return self.probs[:n], self.top1[:n], self.score[:n]

    def stats(self):
        a, b, c, d = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
        check(lib.vg_graph_cache_stats2(self._cache, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), ctypes.byref(d)), 'vg_graph_cache_stats2')
        return {'graphs_captured': a.value, 'graph_launches': b.value, 'graphs_evicted': c.value, 'graphs_live': d.value}


def clip_scores(feat, text_features, stream=None):
    """-> (probs [n,K] f32, top1 [n] int32, top1_score [n] f32), clip_utils.py:42-61."""
    n, dim = feat.shape
    K = text_features.shape[0]
    probs = torch.empty((n, K), dtype=torch.float32, device=feat.device)
    top1 = torch.empty((n,), dtype=torch.int32, device=feat.device)
    score = torch.empty((n,), dtype=torch.float32, device=feat.device)
    if n:
        check(lib.vg_clip_scores(ptr(feat), n, dim, ptr(text_features), K, ptr(probs), ptr(top1), ptr(score),
                                 stream_ptr(stream)), 'vg_clip_scores')
    return probs, top1, score


class ClipWrapper:
    def __init__(self, clip_cfg, model_path, device=None, dtype='f16', synthetic_seed=0):
        assert model_path is not None, 'model_path is None'
        if device is None:
            device = 'cuda'
        self.device = torch.device(device)
        g = clip_cfg.get if hasattr(clip_cfg, 'get') else (lambda k, d=None: getattr(clip_cfg, k, d))
        self.top_k = g('top_k', 1)
        if self.top_k != 1:
            raise NotImplementedError('only top_k = 1 (tools/configs/preprocessor/*.yaml) is implemented on the GPU')
        self.split_size = g('split_size', 50)
        self.template = g('prompt_template')
        class_list = list(g('class_list'))
        self.id_to_class_dict = {idx: name for idx, name in enumerate(class_list)}
        ckpt = os.path.join(str(model_path), str(g('model_name', 'ViT-B-16.pt')))
        if os.path.exists(ckpt):
            weights = clip_weights.load_state_dict(ckpt)
            tf_path = ckpt + '.text_features.npy'
            if not os.path.exists(tf_path):
                # clip_utils.py:22-26: tokenise the prompts, encode_text, normalise -- once, with plain torch, then cached
                from . import clip_text
                prompts = [self.template.format(c) for c in class_list]
                feats = clip_text.text_features(ckpt, prompts, bpe_path=os.environ.get('CLIP_BPE'))
                try:
                    np.save(tf_path, feats)
                except OSError:
                    pass                                    # read-only model directory: recompute next time
            text = torch.from_numpy(np.load(tf_path) if os.path.exists(tf_path) else feats).float()
            self.weights_source = ckpt
        else:
            weights = clip_weights.synthetic_vit_weights(synthetic_seed, **clip_weights.VIT_B16)
            text = clip_weights.synthetic_text_features(synthetic_seed, len(class_list), weights['proj'].shape[1])
            self.weights_source = f'synthetic(seed={synthetic_seed})'
        self.text_features = text.to(self.device).contiguous()
        self.encoder = VitEncoder(weights, dtype=dtype, device=self.device)

    def view(self):
        """The same model for another worker stream: shared weights and text features, own ViT workspace."""
        import copy
        v = copy.copy(self)
        v.encoder = self.encoder.view()
        return v

    def predict_probs(self, crops, stream=None):
        feat = self.encoder.encode(crops, stream)
        return clip_scores(feat, self.text_features, stream)

    def predict_clip_labels(self, crops):
        """crops: [n,3,224,224] CUDA tensor.  Returns (class names, scores) lists of length n, as
        clip_utils.py:49-63 with top_k = 1."""
        probs, top1, score = self.predict_probs(crops)
        top1 = top1.cpu().numpy()
        score = score.cpu().numpy()
        return [self.id_to_class_dict[int(i)] for i in top1], list(score)
