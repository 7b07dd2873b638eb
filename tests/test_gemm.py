"""The projection GEMM of the ViT (csrc/vit.hip k_gemm_f16 / k_gemm_f32) alone, all fused epilogues,
against a plain torch fp32 reference of the same op (floating-point kernel -> torch fp32 reference).
Tolerance: fp16 operands, fp32 accumulate -> |err| <= 2e-3 * sqrt(K) * |x||w| scale; fp32 mode 1e-4."""
import numpy as np
import pytest
import torch


def _ref(X, W, bias, resid, epi):
    acc = X.float() @ W.float().t()
    if epi != 3:
        acc = acc + bias
    if epi == 1:
        acc = acc * torch.sigmoid(1.702 * acc)
    if epi == 2:
        return resid + acc
    return acc


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [1, 0])
@pytest.mark.parametrize('epi', [0, 1, 2, 3])
@pytest.mark.parametrize('shape', [(256, 128, 64), (512, 768, 768), (256, 2304, 768), (256, 768, 3072), (768, 128, 128), (512, 3072, 768), (256, 1536, 96),
                                   (512, 512, 160), (256, 256, 128)])     # K % 64 != 0 -> k_gemm_f16_pp16; two K-tiles -> the shortest pp64 loop
def test_gemm_epilogues(cuda, dtype, epi, shape):
    from vilgod_amd._lib import lib, ptr, stream_ptr, check
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N + K + epi)
    # asymmetric, non-identity operands (a transposed tile write must not pass)
    X = torch.randn(M, K, generator=g) * 0.5
    W = torch.randn(N, K, generator=g) * 0.05
    W[:, 0] += torch.arange(N) * 1e-3
    bias = torch.randn(N, generator=g) * 0.1
    resid = torch.randn(M, N, generator=g)
    td = torch.float16 if dtype == 1 else torch.float32
    Xd, Wd = X.to(td).to(cuda), W.to(td).to(cuda)
    C = torch.zeros(M, N, dtype=torch.float32 if epi == 3 else td, device=cuda)
    R = resid.clone().to(cuda)
    bd = bias.to(cuda)                       # device operands stay referenced until the result has been read back
    check(lib.vg_gemm(dtype, epi, ptr(Xd), ptr(Wd), ptr(bd), ptr(C), ptr(R), M, N, K, stream_ptr()))
    got = (R if epi == 2 else C).float().cpu()
    want = _ref(Xd.cpu(), Wd.cpu(), bias, resid, epi)
    tol = 3e-3 if dtype == 1 else 2e-4
    err = (got - want).abs().max().item()
    assert err < tol * max(1.0, want.abs().max().item()), err


@pytest.mark.gpu
def test_gemm_rejects_bad_shapes(cuda):
    from vilgod_amd._lib import lib, ptr, stream_ptr
    x = torch.zeros(100, 64, dtype=torch.float16, device=cuda)
    assert lib.vg_gemm(1, 0, ptr(x), ptr(x), ptr(x), ptr(x), None, 100, 128, 64, stream_ptr()) == 1


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(512, 768, 768), (256, 768, 3072), (768, 256, 128), (256, 1024, 1024)])
def test_gemm_fp16_residual_epilogue(cuda, shape):
    """epi 4 (fp16 residual stream, in place): resid = f16(resid + f16(X W^T + bias)) -- the arithmetic of the reference's own
    fp16 run (a half Linear output added to a half tensor).  Checked against that formula evaluated from an fp32 product: the two
    roundings are reproduced; the fp32-accumulated product may land one fp16 step away where it sits on a rounding boundary."""
    from vilgod_amd._lib import lib, ptr, stream_ptr, check
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N + K)
    X = (torch.randn(M, K, generator=g) * 0.5).half()
    W = torch.randn(N, K, generator=g) * 0.05
    W[:, 0] += torch.arange(N) * 1e-3
    W = W.half()
    bias = torch.randn(N, generator=g) * 0.1
    resid = (torch.randn(M, N, generator=g) * 2).half()
    R = resid.clone().to(cuda)
    Xd, Wd, bd = X.to(cuda), W.to(cuda), bias.to(cuda)      # stay referenced until the result has been read back
    check(lib.vg_gemm(1, 4, ptr(Xd), ptr(Wd), ptr(bd), None, ptr(R), M, N, K, stream_ptr()))
    proj = (X.float() @ W.float().t() + bias).half()
    want = (resid.float() + proj.float()).half().float()
    got = R.float().cpu()
    ulp = torch.maximum(want.abs(), proj.float().abs()) * 2.0 ** -10 + 1e-4
    # a projection that lands one fp16 step away (accumulation order) moves the sum by that step plus one rounding of the sum
    assert ((got - want).abs() <= 2.5 * ulp).all(), ((got - want).abs() / ulp).max().item()
    assert (got == want).float().mean().item() > 0.9, (got == want).float().mean().item()
    assert lib.vg_gemm(1, 4, ptr(Xd), ptr(Wd), ptr(bd), None, ptr(R), M, 128, K, stream_ptr()) == 1     # N % 256


@pytest.mark.gpu
@pytest.mark.parametrize('epi', [0, 1, 2, 3])
@pytest.mark.parametrize('shape', [(256, 768, 768), (128, 2304, 768), (256, 768, 3072), (384, 128, 16), (128, 3072, 768)])
def test_gemm_f32_on_the_matrix_cores_equals_the_vector_kernel(cuda, epi, shape, monkeypatch):
    """The fp32 parity tower's GEMM on v_mfma_f32_32x32x2_f32 (M % 128 == 0 and N % 128 == 0; round 4) against the 4 x 4 micro-tile
    vector-ALU kernel it replaces (VG_GEMM_F32_MFMA=0): the instruction performs, per output element, two exact fused multiply-adds in
    k order -- the same fmaf chain over ascending k -- so the results are required to be BIT-IDENTICAL, every epilogue."""
    from vilgod_amd._lib import lib, ptr, stream_ptr, check
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N + K + epi)
    X = (torch.randn(M, K, generator=g) * 0.5).to(cuda)
    W = torch.randn(N, K, generator=g) * 0.05
    W[:, 0] += torch.arange(N) * 1e-3
    W = W.to(cuda)
    bias = (torch.randn(N, generator=g) * 0.1).to(cuda)
    resid = torch.randn(M, N, generator=g)
    outs = []
    for mode in ('1', '0'):
        monkeypatch.setenv('VG_GEMM_F32_MFMA', mode)
        C = torch.zeros(M, N, dtype=torch.float32, device=cuda)
        R = resid.clone().to(cuda)
        check(lib.vg_gemm(0, epi, ptr(X), ptr(W), ptr(bias), ptr(C), ptr(R), M, N, K, stream_ptr()))
        torch.cuda.synchronize()
        outs.append((R if epi == 2 else C).clone())
    want = _ref(X.cpu(), W.cpu(), bias.cpu(), resid, epi)
    assert (outs[0].cpu() - want).abs().max().item() < 2e-4 * max(1.0, want.abs().max().item())
    assert torch.equal(outs[0], outs[1])


@pytest.mark.gpu
@pytest.mark.parametrize('K', [768, 3072])
@pytest.mark.parametrize('rows', [257, 260, 300])
def test_gemm_resid_splitk_tail(cuda, K, rows, monkeypatch):
    """The residual GEMMs' split-K tail (vg_gemm_resid_splitk, opt-in through VG_GEMM_SPLITK=8; round 4): with N = 768 a launch of `rows` row tiles is rows x 3 tiles on
    n_cu CUs; the row tiles beyond the last complete round run K-split through scratch.  Against the unsplit launch (VG_GEMM_SPLITK=0):
    the rows of the complete rounds are the SAME numbers, the tail rows agree to fp32 rounding of the partial sums; both against a
    float32 matmul of the same fp16 operands.  300 row tiles leave a last round that is more than half full: not split at all."""
    from vilgod_amd._lib import lib, ptr, stream_ptr, check
    n_cu = torch.cuda.get_device_properties(cuda).multi_processor_count
    M, N = rows * 256, 768
    g = torch.Generator().manual_seed(rows + K)
    X = (torch.randn(M, K, generator=g) * 0.5).half().to(cuda)
    W = torch.randn(N, K, generator=g) * 0.05
    W[:, 0] += torch.arange(N) * 1e-3
    W = W.half().to(cuda)
    bias = (torch.randn(N, generator=g) * 0.1).to(cuda)
    resid = torch.randn(M, N, generator=g).to(cuda)
    scratch = torch.empty(64 << 20, dtype=torch.uint8, device=cuda)
    outs = {}
    monkeypatch.setenv('VG_GEMM_W4', '0')          # the split-K tail belongs to the k_gemm_f16_pp64 family (one tile per workgroup: rounds of n_cu tiles)
    for mode in ('1', '0', 'small'):
        monkeypatch.setenv('VG_GEMM_SPLITK', '0' if mode == '0' else '8')
        R = resid.clone()
        nbytes = 4096 if mode == 'small' else scratch.numel()
        check(lib.vg_gemm_resid_splitk(ptr(X), ptr(W), ptr(bias), ptr(R), M, N, K, ptr(scratch), nbytes, stream_ptr()))
        torch.cuda.synchronize()
        outs[mode] = R
    want = resid + X.float() @ W.float().t() + bias
    scale = max(1.0, want.abs().max().item())
    for mode in outs:
        assert (outs[mode] - want).abs().max().item() < 3e-3 * scale
    assert torch.equal(outs['0'], outs['small'])                    # no room for the partial tiles: the unsplit launch
    total, full = rows * 3, rows * 3 // n_cu
    r_main = full * n_cu // 3
    tail = (rows - r_main) * 3
    split = full >= 1 and 0 < tail <= n_cu // 2 and K >= 1536          # (short K loops are never split: the two extra launches cost as much)
    if not split:
        assert torch.equal(outs['1'], outs['0'])
        return
    assert torch.equal(outs['1'][:r_main * 256], outs['0'][:r_main * 256])
    d = (outs['1'][r_main * 256:] - outs['0'][r_main * 256:]).abs().max().item()
    print(f'{rows} row tiles, K {K}: {tail} tail tiles split; max |split - unsplit| = {d:.2e} (values up to {scale:.1f})')
    assert 0 < d < 2e-5 * scale


@pytest.mark.gpu
@pytest.mark.parametrize('epi', [0, 1, 2, 3])
@pytest.mark.parametrize('shape', [(256, 256, 256), (512, 768, 320), (256, 2304, 768), (768, 768, 3072), (25600, 768, 256), (66560, 256, 256)])
def test_gemm_w4_equals_pp64_bit_for_bit(cuda, epi, shape, monkeypatch):
    """k_gemm_f16_w4 (round 6: persistent workgroups of four waves, the K loop one generated assembly block, token rows as the MFMA's A
    operand, a row-permuted LDS image of W, epilogues straight from the accumulators, the next tile's first pieces requested in the last
    K iterations) computes the same MFMA products in the same K order as k_gemm_f16_pp64 (VG_GEMM_W4=0) and applies the same epilogue
    expressions: the outputs are the same bits.  Shapes: the shortest K loop it takes (four K-tiles), a K that is not a multiple of 256,
    the ViT-B/16 projections, and launches of 300 / 260 tiles on 256 persistent workgroups (every workgroup's second tile starts from
    prefetched pieces while the first tile's stores are in flight)."""
    from vilgod_amd._lib import lib, ptr, stream_ptr, check
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N + K + epi)
    X = torch.randn(M, K, generator=g).half().to(cuda)
    W = (torch.randn(N, K, generator=g) * 0.05).half().to(cuda)
    bias = torch.randn(N, generator=g).to(cuda)
    resid = torch.randn(M, N, generator=g).to(cuda)
    outs = []
    for w4 in ('1', '0', '1'):
        monkeypatch.setenv('VG_GEMM_W4', w4)
        C = torch.zeros(M, N, dtype=torch.float32 if epi == 3 else torch.float16, device=cuda)
        R = resid.clone()
        check(lib.vg_gemm(1, epi, ptr(X), ptr(W), ptr(bias), ptr(C), ptr(R), M, N, K, stream_ptr()))
        torch.cuda.synchronize()
        outs.append(R if epi == 2 else C)
    assert torch.isfinite(outs[0].float()).all()
    assert torch.equal(outs[0], outs[1]), (outs[0].float() - outs[1].float()).abs().max().item()
    assert torch.equal(outs[0], outs[2])
    if M <= 768:
        want = _ref(X.cpu(), W.cpu(), bias.cpu(), resid.cpu(), epi)
        assert (outs[0].float().cpu() - want.float()).abs().max().item() < 3e-3 * max(1.0, want.abs().max().item())


@pytest.mark.gpu
def test_gemm_w4_repeated_launches_are_deterministic(cuda):
    """Race screen for the hand-placed waits of k_gemm_f16_w4 (a read issued before its DMA piece has landed returns the old LDS bytes
    without any fault): 40 launches of a 780-tile GEMM, K = 768, every result equal to the first."""
    from vilgod_amd._lib import lib, ptr, stream_ptr, check
    M, N, K = 66560, 768, 768
    g = torch.Generator().manual_seed(3)
    X = torch.randn(M, K, generator=g).half().to(cuda)
    W = (torch.randn(N, K, generator=g) * 0.05).half().to(cuda)
    bias = torch.randn(N, generator=g).to(cuda)
    R = torch.zeros(1, device=cuda)
    first = None
    for it in range(40):
        C = torch.full((M, N), float('nan'), dtype=torch.float16, device=cuda)
        check(lib.vg_gemm(1, 0, ptr(X), ptr(W), ptr(bias), ptr(C), ptr(R), M, N, K, stream_ptr()))
        torch.cuda.synchronize()
        if first is None:
            first = C
            assert torch.isfinite(C.float()).all()
        else:
            assert torch.equal(first, C), it
