"""The hand-written wide bf16 GEMM (csrc/wgemm.hip) behind the wide nn.Linear layers (reference model.py:228,255-256,
pairedVAE.py:111-119) against torch's fp32 matmul of the same bf16-rounded operands.  The products of two bf16 numbers
are exact in fp32, so the only difference is the order of the fp32 additions: tolerance 2e-5 x sqrt(K) x max|term|
(stated per case below); the kernel itself is bitwise reproducible run to run, which is checked too."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _operand(rows, k, col, gen, dev):
    """bf16 operand for (rows x k): stored [rows][ld] (col False) or [k][ld] (col True), zero padded to ld."""
    full = torch.randn((rows, k), generator=gen, dtype=torch.float32).to(torch.bfloat16)
    if col:
        ld = (rows + 7) // 8 * 8 + 8                 # a wider leading dimension than needed, on purpose
        st = torch.zeros((k, ld), dtype=torch.bfloat16)
        st[:, :rows] = full.t()
    else:
        ld = (k + 7) // 8 * 8
        st = torch.zeros((rows, ld), dtype=torch.bfloat16)
        st[:, :k] = full
    return full.float().to(dev), st.to(dev), ld


CASES = [
    # M, N, K, a_col, b_col, bias, mask
    (256, 256, 64, 0, 0, False, False),
    (300, 517, 203, 0, 0, True, False),
    (300, 517, 203, 0, 1, False, True),
    (300, 517, 203, 1, 0, True, True),
    (300, 517, 203, 1, 1, False, False),
    (37, 70, 9, 1, 1, True, False),
    (1024, 1000, 4100, 0, 0, True, True),       # runs as several K-ranges (workspace + ordered reduction)
    (1024, 900, 2500, 0, 1, False, True),
    (700, 1300, 1024, 1, 1, False, False),
]


@pytest.mark.parametrize("M,N,K,a_col,b_col,with_bias,with_mask", CASES)
def test_wide_gemm_equals_the_fp32_product_of_the_bf16_operands(M, N, K, a_col, b_col, with_bias, with_mask):
    from scpair_amd import ops, _lib
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(1000 * a_col + 100 * b_col + M + N + K)
    Af, A, lda = _operand(M, K, bool(a_col), gen, dev)
    Bf, B, ldb = _operand(N, K, bool(b_col), gen, dev)
    bias = torch.randn((N,), generator=gen).to(dev) if with_bias else None
    mask = None
    ldm = 0
    if with_mask:
        ldm = (N + 7) // 8 * 8
        mask = (torch.randn((M, ldm), generator=gen) - 0.3).to(torch.bfloat16).to(dev)
    ldc = N + 3
    C = torch.full((M, ldc), 7.0, device=dev)
    ops.gemm_bf16(a_col, b_col, M, N, K, A, lda, B, ldb, C, ldc, bias=bias, mask=mask, ldm=ldm)
    want = Af.double() @ Bf.double().t()
    if bias is not None:
        want = want + bias.double()
    if mask is not None:
        want = torch.where(mask[:, :N].float() > 0, want, torch.zeros_like(want))
    got = C[:, :N].double()
    tol = 2e-5 * (K ** 0.5) * 16.0                      # |a b| <= ~16 for N(0,1) operands
    err = (got - want).abs().max().item()
    assert err <= tol, f"max abs error {err} > {tol}"
    assert torch.all(C[:, N:] == 7.0), "the kernel wrote outside its N columns"
    C2 = torch.full((M, ldc), 7.0, device=dev)
    ops.gemm_bf16(a_col, b_col, M, N, K, A, lda, B, ldb, C2, ldc, bias=bias, mask=mask, ldm=ldm)
    assert torch.equal(C, C2), "not bitwise reproducible"
    if K >= 2048:
        assert _lib.lib().scp_gemm_bf16_nsplit(M, N, K) > 1


def test_wide_gemm_rejects_misaligned_operands():
    from scpair_amd import ops
    dev = torch.device("cuda:0")
    A = torch.zeros((64, 72), dtype=torch.bfloat16, device=dev)
    B = torch.zeros((64, 72), dtype=torch.bfloat16, device=dev)
    C = torch.zeros((64, 64), device=dev)
    with pytest.raises((ValueError, RuntimeError)):
        ops.gemm_bf16(0, 0, 64, 64, 70, A, 70, B, 72, C, 64)        # lda not a multiple of 8
    with pytest.raises((ValueError, RuntimeError)):
        ops.gemm_bf16(0, 0, 64, 64, 80, A, 72, B, 72, C, 64)        # lda smaller than K


def test_wide_gemm_output_of_two_gib_and_more():
    """An output of 2 GiB or more cannot go through 32-bit buffer offsets: edge and masked tiles then take the guarded
    64-bit stores.  27 000 x 20 000 fp32 (2.16 GB, ragged in both dimensions) with bias and mask, short K."""
    from scpair_amd import ops
    dev = torch.device("cuda:0")
    M, N, K = 27000, 20000, 64
    gen = torch.Generator(device=dev).manual_seed(11)
    A = torch.randn((M, K), device=dev, generator=gen).to(torch.bfloat16)
    B = torch.randn((N, K), device=dev, generator=gen).to(torch.bfloat16)
    bias = torch.randn((N,), device=dev, generator=gen)
    mask = (torch.randn((M, N), device=dev, generator=gen) - 0.3).to(torch.bfloat16)
    C = torch.full((M, N), 7.0, device=dev)
    assert C.numel() * 4 >= 1 << 31
    ops.gemm_bf16(0, 0, M, N, K, A, K, B, K, C, N, bias=bias, mask=mask, ldm=N)
    want = A.float() @ B.float().t() + bias
    want = torch.where(mask.float() > 0, want, torch.zeros_like(want))
    assert float((C - want).abs().max()) <= 2e-5 * 8 * 16.0
    del want
    C2 = torch.full((M, N), 7.0, device=dev)
    ops.gemm_bf16(0, 0, M, N, K, A, K, B, K, C2, N)          # no mask: interior tiles plain stores, edge tiles guarded
    want = A.float() @ B.float().t()
    assert float((C2 - want).abs().max()) <= 2e-5 * 8 * 16.0


def test_wide_gemm_output_just_below_two_gib():
    """C ends 1.1 MB below 2^31 with ragged M and N: the corner tile's rows straddle M and its columns run past N.  An
    out-of-range mark OR-ed into a 32-bit base offset would be carried out again by the row term there and the dropped
    stores would land in the first rows of C instead (26 830 x 20 000 fp32 = 2 146 400 000 bytes)."""
    from scpair_amd import ops
    dev = torch.device("cuda:0")
    M, N, K = 26830, 20000, 64
    assert (1 << 31) - 28 * N * 4 < M * N * 4 < (1 << 31)
    gen = torch.Generator(device=dev).manual_seed(12)
    A = torch.randn((M, K), device=dev, generator=gen).to(torch.bfloat16)
    B = torch.randn((N, K), device=dev, generator=gen).to(torch.bfloat16)
    bias = torch.randn((N,), device=dev, generator=gen)
    for with_mask in (False, True):
        mask = (torch.randn((M, N), device=dev, generator=gen) - 0.3).to(torch.bfloat16) if with_mask else None
        C = torch.full((M, N), 7.0, device=dev)
        ops.gemm_bf16(0, 0, M, N, K, A, K, B, K, C, N, bias=bias, mask=mask, ldm=N if with_mask else 0)
        want = A.float() @ B.float().t() + bias
        if with_mask:
            want = torch.where(mask.float() > 0, want, torch.zeros_like(want))
        assert float((C[:64] - want[:64]).abs().max()) <= 2e-5 * 8 * 16.0, "the first rows of C were overwritten"
        assert float((C - want).abs().max()) <= 2e-5 * 8 * 16.0
        del want, C, mask
