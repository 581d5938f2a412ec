"""The "f32x3" mode (option f32_split, csrc/gemm.hip SplitCfg): f32 operands in HBM, every matrix product as three bf16 MFMAs over
(hi, lo) operand pairs.  Kernel level: the f32 cases of tests/test_ops_gpu.py re-run with the option on, against the same plain
PyTorch fp32 expressions, at 1e-4 of the output's range (16 significand bits per operand; the exact-f32 kernels sit at 2e-5);
plus direct error measurements against float64 that show the option is live and how far it is from the exact kernels."""
import pytest
import torch

from tests import test_ops_gpu as T

pytestmark = pytest.mark.gpu

F32 = torch.float32
SPLIT_TOL = 1e-4


@pytest.fixture(autouse=True)
def _seed():
    torch.manual_seed(4321)


@pytest.fixture()
def split(monkeypatch):
    from emoasr_amd import ops
    monkeypatch.setattr(T, "_tol", lambda dtype, f32=2e-5, bf16=2e-2: max(f32, SPLIT_TOL) if dtype == F32 else bf16)
    ops.split_products(True)
    yield
    ops.split_products(False)


@pytest.fixture()
def tr_mode():
    return 1


@pytest.mark.parametrize("M,N,K", [(300, 256, 256), (1000, 1024, 256), (77, 1000, 256), (130, 256, 4864), (4100, 256, 1024),
                                   (64, 64, 32), (9000, 256, 256)])
def test_split_gemm_nt(dev, split, M, N, K):
    T.test_gemm_nt_plain(dev, F32, M, N, K)


def test_split_gemm_nt_epilogue(dev, split):
    T.test_gemm_nt_epilogue(dev, F32)


@pytest.mark.parametrize("K,N1,N2", [(1000, 256, 256), (3001, 1024, 256), (777, 1000, 256), (500, 256, 2304), (64, 64, 64)])
def test_split_gemm_tn(dev, split, tr_mode, K, N1, N2):
    T.test_gemm_tn(dev, F32, tr_mode, K, N1, N2)


def test_split_gemm_nn_and_batched(dev, split, tr_mode):
    T.test_gemm_nn(dev, F32, tr_mode)


def test_split_gemm_tn_grouped(dev, split, tr_mode):
    T.test_gemm_tn_grouped(dev, F32, tr_mode)


@pytest.mark.parametrize("Tn,Fd", [(67, 80), (70, 83), (9, 7)], ids=["odd-T1", "even-T1-F1", "tiny"])
def test_split_frontend(dev, split, tr_mode, Tn, Fd):
    T.test_frontend(dev, F32, tr_mode, Tn, Fd)


def test_split_is_live_and_sixteen_bits_wide(dev):
    """against float64: the split product's error is ~2^-17 of the operands' product scale -- far below bf16's 2^-9, above the exact
    kernel's ~2^-24 -- and it changes the bits (the option is not a no-op)"""
    from emoasr_amd import ops
    M, N, K = 2000, 512, 1024
    a, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * K ** -0.5
    ref = a.double() @ b.double().t()
    scale = ref.abs().max().item()
    exact = ops.gemm_nt(a, b)
    ops.split_products(True)
    try:
        got = ops.gemm_nt(a, b)
    finally:
        ops.split_products(False)
    e_exact = (exact.double() - ref).abs().max().item() / scale
    e_split = (got.double() - ref).abs().max().item() / scale
    e_bf16 = (ops.gemm_nt(a.bfloat16(), b.bfloat16()).double() - ref).abs().max().item() / scale
    assert not torch.equal(got, exact)
    assert e_exact < 2e-6 and e_split < 2e-5 and e_bf16 > 30 * e_split, (e_exact, e_split, e_bf16)


@pytest.mark.parametrize("case", ["rel", "rel_long", "plain_mask", "causal", "cross"])
@pytest.mark.parametrize("mat", ["stored", True, False], ids=["stored", "gemmbwd", "recompute"])
def test_split_attention(dev, split, tr_mode, case, mat):
    """forward kernel and the score-recomputing dQ kernel with split products (Mma<f32s>); "gemmbwd" is the training path's form
    (dQ kernel + the dV / dK / table-gradient products as split GEMMs); the other two keep the exact kernels behind a split forward"""
    T.test_attention(dev, F32, tr_mode, case, mat)


def test_split_attention_dropout(dev, split):
    T.test_attention_dropout(dev)
