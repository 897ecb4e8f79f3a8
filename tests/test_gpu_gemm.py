"""The split-bf16 MFMA GEMM of the recognition network (csrc/gemm_bf16.hip through stove_gemm_bf16) against fp64 matmul:
all four operand layouts, ragged M / N / K (predicated tile edges), split-K, bias, and exact-integer data that pins the
fragment / transposed-read / k-permutation maps (a symmetric or random check can hide a transposed tile)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


def _ref(a, b, bias, ak, bk):
    A = a.double().t() if ak else a.double()
    B = b.double().t() if bk else b.double()
    c = A @ B.t()
    return c + bias.double() if bias is not None else c


@pytest.mark.parametrize('tile', [0, 3])          # 0: chosen from the shape (256 x 128 / 128 x 128); 3: 256 x 256, eight waves of 64 x 128
@pytest.mark.parametrize('ak,bk', [(False, False), (False, True), (True, True), (True, False)])
@pytest.mark.parametrize('M,N,K,splitk', [(256, 128, 32, 1), (512, 256, 96, 1), (300, 132, 100, 1), (20, 1024, 1024, 1),
                                            (1024, 256, 3000, 8), (64, 64, 4, 3), (260, 256, 20000, 32)])
def test_gemm_layouts_and_edges(ak, bk, M, N, K, splitk, tile):
    from stove_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a = torch.randn((K, M) if ak else (M, K), generator=g).to(DEV)
    b = torch.randn((K, N) if bk else (N, K), generator=g).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV) if splitk == 1 else None
    ref = _ref(a, b, bias, ak, bk)
    scale = float(ref.abs().max())
    for nsplit, tol in ((2, 1.5e-5), (1, 2e-2)):
        c = ops.gemm_bf16(a, b, bias, ak, bk, nsplit, splitk, tile=tile)
        err = float((c.double() - ref).abs().max()) / scale
        assert err < tol, (nsplit, err)


@pytest.mark.parametrize('tile', [0, 3])
@pytest.mark.parametrize('ak,bk', [(False, False), (False, True), (True, True), (True, False)])
def test_gemm_exact_on_small_integers(ak, bk, tile):
    """bf16 holds integers up to 256 exactly and fp32 accumulation of such products is exact: any wrong lane / k map shows
    as a non-zero difference.  B is asymmetric (not a function of |row - col| or row + col)."""
    from stove_amd import ops
    M, N, K = 256 + 16, (256 if tile >= 3 else 128) + 4, 64 + 8
    m, n, k = torch.arange(M).view(-1, 1), torch.arange(N).view(-1, 1), torch.arange(K).view(1, -1)
    a = ((m * 3 + k * 5) % 17 - 8).float()
    b = ((n * 7 + k * 11 + (n * k) % 5) % 13 - 6).float()
    at, bt = (a.t().contiguous() if ak else a), (b.t().contiguous() if bk else b)
    for nsplit in (1, 2):
        c = ops.gemm_bf16(at.to(DEV), bt.to(DEV), None, ak, bk, nsplit, 1, tile=tile)
        assert torch.equal(c.cpu(), a @ b.t())


@pytest.mark.parametrize('tile', [0, 1, 2, 3])
@pytest.mark.parametrize('M,N,K,splitk', [(256, 128, 32, 1), (300, 132, 100, 1), (20, 1024, 1024, 1), (1024, 256, 3000, 8), (3200, 1024, 1024, None)])
def test_gemm_half_pieces(M, N, K, splitk, tile):
    """nsplit 3 (csrc/split16.h): IEEE-half hi + lo pieces carry 22 bits of each operand; the product is fp32-grade -- here held to
    1e-6 of the largest entry (measured 1-4e-7), an order below the bf16 pieces' 4.5e-6 -- with the same tiles, ragged edges and split-K."""
    from stove_amd import ops
    g = torch.Generator().manual_seed(M + N * 5 + K * 3)
    a = torch.randn(M, K, generator=g).to(DEV)
    b = torch.randn(N, K, generator=g).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV) if splitk == 1 else None
    ref = _ref(a, b, bias, False, False)
    c = ops.gemm_bf16(a, b, bias, False, False, 3, splitk, tile=tile)
    err = float((c.double() - ref).abs().max() / ref.abs().max())
    assert err < 1e-6, err


def test_gemm_half_pieces_exact_integers_and_layout_refusal():
    """Integers below 2048 are exact in one half piece: any wrong lane / k map shows as a non-zero difference.  The half pieces exist for
    row-major operands only; the other layouts are refused, not silently computed on bf16."""
    from stove_amd import ops
    M, N, K = 272, 132, 72
    m, n, k = torch.arange(M).view(-1, 1), torch.arange(N).view(-1, 1), torch.arange(K).view(1, -1)
    a = ((m * 3 + k * 5) % 17 - 8).float()
    b = ((n * 7 + k * 11 + (n * k) % 5) % 13 - 6).float()
    for tile in (1, 2, 3):
        assert torch.equal(ops.gemm_bf16(a.to(DEV), b.to(DEV), None, False, False, 3, 1, tile=tile).cpu(), a @ b.t())
    for ak, bk in ((True, False), (False, True), (True, True)):
        with pytest.raises(RuntimeError):
            ops.gemm_bf16((a.t().contiguous() if ak else a).to(DEV), (b.t().contiguous() if bk else b).to(DEV), None, ak, bk, 3, 1)


@pytest.mark.parametrize('scale_a,scale_b', [(1.0, 1.0), (1.0, 3e-2), (1.0, 3e-3), (1.0, 1e-4), (0.1, 0.1), (1e-2, 1.0), (1e-3, 1.0), (50.0, 50.0), (1e3, 1e-3)])
def test_gemm_half_pieces_range(scale_a, scale_b):
    """The domain of nsplit 3, written down (csrc/split16.h): an element keeps max(2^-22 |x|, 2^-25) -- the lo piece, at most 2^-11 |x|,
    is a subnormal half once |x| < 1/4 -- and must stay below 65504.  B (the weights of the forward products) is cut as 2^8 B, so
    its floor is 2^-33 and its ceiling 256; A (frames in [0, 1], hidden states in (-1, 1)) is taken as it is.  Bound: 1e-6 of the
    largest entry (fp32 accumulation included) plus the floors' sums, sqrt(K) x floor x the other operand's largest element."""
    from stove_amd import ops
    M, N, K = 512, 256, 1024
    g = torch.Generator().manual_seed(11)
    a = (torch.randn(M, K, generator=g) * scale_a).to(DEV)
    b = (torch.randn(N, K, generator=g) * scale_b).to(DEV)
    ref = _ref(a, b, None, False, False)
    c = ops.gemm_bf16(a, b, None, False, False, 3, 1)
    assert bool(torch.isfinite(c).all())
    err = float((c.double() - ref).abs().max())
    floor = K ** 0.5 * (2.0 ** -25 * float(b.abs().max()) + 2.0 ** -33 * float(a.abs().max()))
    assert err < 1e-6 * float(ref.abs().max()) + floor, (err, float(ref.abs().max()), floor)
    if scale_a >= 1.0:          # A at O(1), B anywhere above its (shifted) floor: fp32-grade whatever the weights' scale
        assert err < 1e-6 * float(ref.abs().max()), (err, float(ref.abs().max()))


def test_gemm_encoder_shapes_accuracy():
    """The headline shapes (25 600 frames): input projection x W_ih^T and its weight gradient dgx^T x; error of the 3-MFMA
    split against fp64, next to the fp32 library GEMM's own error."""
    from stove_amd import ops
    g = torch.Generator().manual_seed(0)
    x = torch.rand(25600, 1024, generator=g).to(DEV)
    w = (torch.randn(1024, 1024, generator=g) * 0.03).to(DEV)
    ref = x[:2048].double() @ w.double().t()
    got = ops.gemm_bf16(x, w)[:2048].double()
    lib = (x[:2048] @ w.t()).double()
    e_split = float((got - ref).abs().max() / ref.abs().max())
    e_lib = float((lib - ref).abs().max() / ref.abs().max())
    assert e_split < 1e-5, (e_split, e_lib)
    dg = torch.randn(25600, 1024, generator=g).to(DEV)
    ref = dg.double().t() @ x.double()
    got = ops.gemm_bf16(dg, x, None, True, True, 2, 8).double()
    assert float((got - ref).abs().max() / ref.abs().max()) < 1e-5


def test_gemm_unaligned_fc1_shapes():
    """fc1 of the recognition network has 50 columns (reference encoder.py:25): C with N = ldc = 50 (element-wise epilogue
    with bias), A with lda = K = 50 and A^T with M = ld = 50 (element-wise operand loads), also on exact integers and with
    an 8-byte-aligned base pointer."""
    from stove_amd import ops
    g = torch.Generator().manual_seed(5)
    rows = 3 * 700 + 1
    h = torch.randn(rows, 256, generator=g).to(DEV)
    w1 = torch.randn(50, 256, generator=g).to(DEV)
    b1 = torch.randn(50, generator=g).to(DEV)
    d = torch.randn(rows, 50, generator=g).to(DEV)
    for nsplit, tol in ((2, 1.5e-5), (1, 2e-2)):
        for tile in (1, 2):
            a1 = ops.gemm_bf16(h, w1, bias=b1, nsplit=nsplit, tile=tile)
            ref = h.double() @ w1.double().t() + b1.double()
            assert a1.shape == (rows, 50) and float((a1.double() - ref).abs().max() / ref.abs().max()) < tol
        gh = ops.gemm_bf16(d, w1, b_kmajor=True, nsplit=nsplit, splitk=1, tile=2)
        ref = d.double() @ w1.double()
        assert float((gh.double() - ref).abs().max() / ref.abs().max()) < tol
        for splitk in (1, 4, None):
            gw = ops.gemm_bf16(d, h, a_kmajor=True, b_kmajor=True, nsplit=nsplit, splitk=splitk, tile=2)
            ref = d.double().t() @ h.double()
            assert gw.shape == (50, 256) and float((gw.double() - ref).abs().max() / ref.abs().max()) < tol
    # exact integers through the element-wise loads, base pointer off the 16-byte grid
    m, k, n = torch.arange(131).view(-1, 1), torch.arange(50).view(1, -1), torch.arange(256).view(-1, 1)
    a = ((m * 3 + k * 5) % 17 - 8).float()
    b = ((n * 7 + k * 11 + (n * k) % 5) % 13 - 6).float()
    buf = torch.zeros(a.numel() + 2, device=DEV)
    a_off = buf[2:].view(131, 50)
    a_off.copy_(a)
    c = ops.gemm_bf16(a_off, b.t().contiguous().to(DEV), b_kmajor=True, nsplit=2, splitk=1, tile=2)
    assert torch.equal(c.cpu(), a @ b.t())
    at = torch.zeros(50 * 131 + 2, device=DEV)[2:].view(50, 131)
    at.copy_(a.t())
    c = ops.gemm_bf16(at, b.t().contiguous().to(DEV), a_kmajor=True, b_kmajor=True, nsplit=1, splitk=1, tile=2)
    assert torch.equal(c.cpu(), a @ b.t())


def test_gemm_splitk_bias_and_balanced_rows():
    """A bias with split-K (added by the slice sum) and the row-balanced input projection of the recognition network (complete
    rounds of 256 x 128 tiles + a split-K launch for the rows of the incomplete round) against fp64."""
    from stove_amd import ops
    g = torch.Generator().manual_seed(3)
    a = torch.randn(1024, 512, generator=g).to(DEV)
    b = torch.randn(256, 512, generator=g).to(DEV)
    bias = torch.randn(256, generator=g).to(DEV)
    ref = a.double() @ b.double().t() + bias.double()
    got = ops.gemm_bf16(a, b, bias=bias, nsplit=2, splitk=8)
    assert float((got.double() - ref).abs().max() / ref.abs().max()) < 1.5e-5
    cus = torch.cuda.get_device_properties(DEV).multi_processor_count
    n = (3 * cus // 8) * 256 + 4 * 256 + 37                 # three complete rounds of tiles + an incomplete one with a ragged end
    x = torch.rand(n, 1024, generator=g).to(DEV)
    w = (torch.randn(1024, 1024, generator=g) * 0.03).to(DEV)
    bb = torch.randn(1024, generator=g).to(DEV)
    got = ops._gemm_rows_balanced(x, w, bb, 2)
    plain = ops.gemm_bf16(x, w, bias=bb, nsplit=2, splitk=1)
    sel = torch.cat([torch.arange(0, 512), torch.arange(n - 1400, n)]).to(DEV)
    ref = x[sel].double() @ w.double().t() + bb.double()
    assert got.shape == plain.shape == (n, 1024)
    assert float((got[sel].double() - ref).abs().max() / ref.abs().max()) < 1e-5
    assert float((got - plain).abs().max() / plain.abs().max()) < 1e-5
