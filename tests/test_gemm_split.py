"""The opt-in bf16-split GEMM experiment (csrc/gemm_split.hip): fp32 operands split into 2 / 3
bf16 planes, 3 / 6 bf16 MFMA products per fp32 product, fp32 accumulation.  Checked against float64 and against the product's
fp32-MFMA kernel on the same inputs: the 6-term form must be fp32-grade (error within 1.5x the fp32 kernel's), the 3-term form is
reported with the factor it really has (its dropped lo x lo term is 2^-16 of a product)."""
import pytest
import torch


def _run(dev, shapes):
    from omni3d_amd.kernels import wino
    g = torch.Generator().manual_seed(4)
    out = []
    for B, M, K, C in shapes:
        V = torch.randn(B, M, C, generator=g).to(dev)
        U = (torch.randn(B, K, C, generator=g) * 0.1).to(dev)
        ref = torch.bmm(V.double().cpu(), U.double().cpu().transpose(1, 2))
        scale = float(ref.abs().max())
        e32 = float((wino.gemm_batched(V, U).double().cpu() - ref).abs().max()) / scale
        e6 = float((wino.gemm_batched_split(V, U, 6).double().cpu() - ref).abs().max()) / scale
        e3 = float((wino.gemm_batched_split(V, U, 3).double().cpu() - ref).abs().max()) / scale
        out.append((e32, e6, e3))
        assert e6 <= 1.5 * e32 + 1e-7, (B, M, K, C, e32, e6)
        assert e3 <= 2e-5 and e3 >= e6 * 0.5, (e32, e6, e3)          # 2^-16-grade products, never better than the 6-term form by luck alone
    return out


def test_gemm_split_emulated(emu_lib):
    _run("cpu", [(2, 70, 40, 64), (1, 128, 128, 32), (3, 33, 129, 96)])


@pytest.mark.gpu
def test_gemm_split_gpu(hip_lib):
    errs = _run("cuda", [(36, 4096, 256, 256), (36, 1024, 128, 128), (2, 70, 40, 64), (3, 33, 129, 96)])
    print("max |err| / max |ref|  (fp32 MFMA, 6-term split, 3-term split):", errs)
