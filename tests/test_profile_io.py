"""omni3d_amd/profile_io.py: the executed-flop counter of the bench line knows every MFMA launcher of the C ABI."""
import ctypes

from omni3d_amd import profile_io as P


def test_every_mfma_entry_point_of_the_header_is_classified():
    sig = P.abi_signatures()
    assert len(sig) >= 115
    hits = [n for n in sig if P.MFMA_ENTRY.search(n) and not n.startswith("omni_resize")]
    assert len(hits) >= 30
    unknown = [n for n in hits if P.flop_class(n) is None]
    assert not unknown, unknown
    assert all(P.flop_class(n) is None for n in sig if not P.MFMA_ENTRY.search(n))
    assert P.VALU_ENTRIES <= set(sig)          # the exclusions name real entries


def test_flop_formulas():
    # direct convolution, every spelling: 2 N OH OW K R S C
    want = 2.0 * 4 * 64 * 64 * 128 * 9 * 64
    sig = P.abi_signatures()
    for name in ("omni_conv2d_fwd", "omni_conv2d_fwd_algo", "omni_conv2d_fwd_det", "omni_conv2d_fwd_stats", "omni_conv2d_dgrad", "omni_conv2d_dgrad_det",
                 "omni_conv2d_wgrad_algo", "omni_conv2d_wgrad_det"):
        vals = {"N": 4, "H": 128, "W": 128, "C": 64, "K": 128, "R": 3, "S": 3, "stride": 2, "pad": 1}
        a = [vals.get(n) for n in sig[name]]            # pointers, pitches, plan: None
        assert P.executed_flops(name, a) == want, name
    a = [{"N": 4, "H": 128, "W": 128, "C": 64, "K": 128, "R": 3, "S": 3, "stride": 2, "pad": 1, "plan": 1234}.get(n) for n in sig["omni_conv2d_fwd_det"]]
    assert P.executed_flops("omni_conv2d_fwd_det", a) == 0.0       # planning call: nothing is launched
    a = [{"N": 4, "H": 512, "W": 512, "C": 16, "K": 32, "R": 3}.get(n) for n in sig["omni_stem_conv_s2_dgrad"]]
    assert P.executed_flops("omni_stem_conv_s2_dgrad", a) == 2.0 * 4 * 256 * 256 * 32 * 9 * 16
    a = [{"batch": 36, "M": 4096, "C": 256, "K": 256}.get(n) for n in sig["omni_gemm_batched_wgrad_det"]]
    assert P.executed_flops("omni_gemm_batched_wgrad_det", a) == 2.0 * 36 * 4096 * 256 * 256
    a = [{"batch": 1, "M": 2048, "N": 1024, "K": 12544}.get(n) for n in sig["omni_gemm_engine_det"]]
    assert P.executed_flops("omni_gemm_engine_det", a) == 2.0 * 2048 * 1024 * 12544
    # host arrays of the multi-problem launch and of the fused RPN head
    n = 3
    ints = lambda xs: ctypes.cast((ctypes.c_int * n)(*xs), ctypes.c_void_p)      # noqa: E731
    keep = [ints([36, 16, 36]), ints([1024, 256, 256]), ints([128, 512, 256]), ints([128, 512, 256])]
    a = [dict(zip(("batch", "M", "C", "K"), keep), n=n).get(k) for k in sig["omni_gemm_batched_wgrad_multi"]]
    assert P.executed_flops("omni_gemm_batched_wgrad_multi", a) == 2.0 * (36 * 1024 * 128 * 128 + 16 * 256 * 512 * 512 + 36 * 256 * 256 * 256)
    px = (ctypes.c_longlong * 2)(65536, 16384)
    a = [{"pix": ctypes.cast(px, ctypes.c_void_p), "nlev": 2}.get(k) for k in sig["omni_rpn_head16_fwd"]]
    assert P.executed_flops("omni_rpn_head16_fwd", a) == 2.0 * (65536 + 16384) * 256 * 16
    assert P.executed_flops("omni_rpn_head16_dgrad", a) == 0.0 and P.executed_flops("omni_bn_fwd_algo", []) == 0.0
