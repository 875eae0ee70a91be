"""CPU: the C-ABI shared library builds for gfx950, loads, and exports every symbol that
include/dsr_hip.h declares, with the argument counts the ctypes binding assumes.  No compute calls."""
import ctypes
import importlib
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "deep-super-resolution_amd"


@pytest.fixture(scope="module")
def so():
    b = importlib.import_module(PKG + "._build")
    return b.build()


def declared():
    src = open(os.path.join(ROOT, "include", "dsr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|size_t|const char\*)\s+(dsr_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return out


def test_header_symbols_exported(so):
    lib = ctypes.CDLL(so)
    decl = declared()
    assert len(decl) >= 30
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in include/dsr_hip.h but not exported"


def test_ctypes_signatures_match_header(so):
    L = importlib.import_module(PKG + "._lib")
    decl = declared()
    assert set(L.SIGNATURES) == set(decl), set(L.SIGNATURES) ^ set(decl)
    for name, (_, args) in L.SIGNATURES.items():
        assert len(args) == decl[name], (name, len(args), decl[name])
    L.lib()
    assert L.lib().dsr_abi_version() == L.ABI_VERSION == 7


def test_host_side_descriptor_checks(so):
    """Argument validation runs on the host before any launch, so it is testable without a GPU."""
    L = importlib.import_module(PKG + "._lib")
    lib = L.lib()
    oh, ow = ctypes.c_int(), ctypes.c_int()
    d = L.ConvDesc(L.BF16, 2, 16, 16, 64, 64, 3, 3, 2, 1, 0)
    assert lib.dsr_conv_out_size(ctypes.byref(d), ctypes.byref(oh), ctypes.byref(ow)) == 0
    assert (oh.value, ow.value) == (8, 8)
    assert lib.dsr_conv_stats_rows(ctypes.byref(d)) == 1
    assert lib.dsr_conv_packed_elems(ctypes.byref(d), 0) == 9 * 64 * 64
    bad = L.ConvDesc(L.BF16, 2, 16, 16, 64, 64, 11, 11, 1, 5, 0)      # 121 taps > 96
    assert lib.dsr_conv_out_size(ctypes.byref(bad), ctypes.byref(oh), ctypes.byref(ow)) < 0
    assert b"taps" in lib.dsr_last_error()
    refl = L.ConvDesc(L.BF16, 1, 1, 1, 8, 8, 3, 3, 1, 1, 1)           # reflect pad >= size: torch raises too
    assert lib.dsr_conv_out_size(ctypes.byref(refl), ctypes.byref(oh), ctypes.byref(ow)) < 0
    with pytest.raises(RuntimeError):
        L.check(-1)


def test_bad_arguments_return_codes_not_crashes(so):
    """Round-1 bring-up crashes, restated as contract tests.  Every entry point validates on the host and
    returns a negative code BEFORE anything is launched: a null device pointer would otherwise become a GPU memory fault
    (the runtime aborts the process: the 07:11 abort in test_conv_act_fwd_bwd came from a partial-sum buffer that was too
    small for the compaction pass behind it), a null HOST table a segfault.  None of these calls needs a GPU."""
    L = importlib.import_module(PKG + "._lib")
    lib = L.lib()
    N = None
    st = None
    one = ctypes.c_void_p(16)            # a non-null "pointer" that is never dereferenced: validation fails first
    d = L.ConvDesc(L.BF16, 2, 16, 16, 64, 64, 3, 3, 1, 1, 0)
    ep = L.Epilogue(0, 0.0, N, N, N, 0, N, N, N, N)
    calls = [
        lambda: lib.dsr_conv_out_size(N, N, N),
        lambda: lib.dsr_conv_out_size(ctypes.byref(d), N, N),
        lambda: lib.dsr_conv_fwd(ctypes.byref(d), N, N, ctypes.byref(ep), N, st),
        lambda: lib.dsr_conv_fwd(N, one, one, ctypes.byref(ep), one, st),
        lambda: lib.dsr_conv_dgrad(ctypes.byref(d), N, N, N, N, 0, st),
        lambda: lib.dsr_conv_wgrad(ctypes.byref(d), N, N, N, N, 0, st),
        lambda: lib.dsr_conv_pack_weight(ctypes.byref(d), N, N, N, st),
        lambda: lib.dsr_conv_pack_weight_multi(0, 3, N, N, N, N, N, N, st),
        lambda: lib.dsr_conv_first_bwd(ctypes.byref(d), N, N, N, 1, 0.2, N, N, N, 0, st),
        lambda: lib.dsr_conv_first_bwd_recompute(ctypes.byref(d), N, N, N, N, 1, 0.2, N, N, N, 0, st),
        lambda: lib.dsr_conv_dgrad_first_bwd(ctypes.byref(d), ctypes.byref(d), N, N, N, N, N, 1, 0.2, N, N, N, 0, st),   # not a layer pair
        lambda: lib.dsr_conv_dgrad_bn(ctypes.byref(d), N, N, N, N, N, N, 1, 0.2, N, st),          # stride 1
        lambda: lib.dsr_conv_dgrad_ps(ctypes.byref(d), N, N, N, N, N, N, st),                        # not the 9x9 tail
        lambda: lib.dsr_pw_nchw_to_nhwc(0, N, N, 1, 3, 4, 4, 8, st),
        lambda: lib.dsr_pw_nhwc_to_nchw(0, one, one, 1, 3, 4, 4, 7, st),          # Cp % 8
        lambda: lib.dsr_pw_sum_rows(N, 4, 8, 0, 8, 1.0, N, 0, 1, st),
        lambda: lib.dsr_pw_bn_finalize(N, 4, 64, 64, 64, 16.0, N, N, N, N, N, 0.1, 1e-5, 1, N, N, N, N, st),
        lambda: lib.dsr_pw_bn_eval_affine(N, N, N, N, 1e-5, 64, 64, N, N, N, N, st),
        lambda: lib.dsr_pw_reduce_blocks(1024, N),
        lambda: lib.dsr_pw_channel_stats(0, N, 64, 64, 1, 64, N, st),
        lambda: lib.dsr_pw_bn_act_fwd(0, N, N, N, N, N, 64, 64, 0, 0.0, N, st),
        lambda: lib.dsr_pw_bn_act_fwd(0, one, one, one, N, one, 64, 4096, 0, 0.0, N, st),   # Cp > 2048: no pixel row per block
        lambda: lib.dsr_pw_bn_act_fwd(2, one, one, one, N, one, 64, 64, 0, 0.0, N, st),     # dtype
        lambda: lib.dsr_pw_bn_act_fwd(0, one, one, one, N, one, 64, 64, L.ACT_PRELU, 0.0, N, st),   # PReLU without its weight
        lambda: lib.dsr_pw_bn_act_bwd_reduce(0, N, N, N, N, N, N, 64, 64, 1, 64, 0, 0.0, N, N, st),
        lambda: lib.dsr_pw_bn_bwd_finalize(N, 1, 64, 64, 64.0, N, N, N, N, N, N, N, st),
        lambda: lib.dsr_pw_bn_act_bwd_apply(0, N, N, N, N, N, N, N, N, N, 64, 64, 0, 0.0, N, 1, st),
        lambda: lib.dsr_pw_act_bwd(0, N, N, N, 1, 4, 4, 64, 64, 0, 1, 0.2, N, 1, 64, N, st),
        lambda: lib.dsr_pw_act_bwd(0, one, one, one, 1, 4, 4, 64, 64, 0, L.ACT_LEAKY, -0.1, N, 1, 64, N, st),   # slope <= 0
        lambda: lib.dsr_pw_act_bwd(0, one, one, one, 1, 4, 4, 64, 64, 0, L.ACT_LEAKY, 0.0, N, 1, 64, N, st),
        lambda: lib.dsr_pw_act_bwd(0, one, one, one, 1, 4, 4, 64, 64, 0, L.ACT_PRELU, 0.0, N, 1, 64, N, st),    # PReLU, no weight
        lambda: lib.dsr_pw_act_bwd_nchw(0, N, N, N, 1, 3, 4, 4, 8, 4, st),
        lambda: lib.dsr_pw_colsum(0, N, 64, 64, 1, 64, N, st),
        lambda: lib.dsr_pw_add(0, N, N, N, 8, st),
        lambda: lib.dsr_pw_axpby_f32(N, N, 1.0, 1.0, N, N, 4, st),
        lambda: lib.dsr_pw_diff_loss(N, N, N, 16, 0, N, 1, st),
        lambda: lib.dsr_pw_diff_loss(one, one, one, 16, 7, one, 1, st),           # mode
        lambda: lib.dsr_pw_bce_const(N, 4, 1.0, N, N, 0, st),
        lambda: lib.dsr_pw_adam(N, N, N, N, 16, 1e-3, 0.9, 0.999, 1e-8, N, 1.0, N, st),
        lambda: lib.dsr_pw_adam_multi(2, N, N, N, N, N, 1e-3, 0.9, 0.999, 1e-8, N, 1.0, st),
        lambda: lib.dsr_pw_incr(N, st),
        lambda: lib.dsr_cast16(0, N, N, 16, st),
        lambda: lib.dsr_flatten(0, N, N, 2, 4, 8, 8, 0, 0, st),
        lambda: lib.dsr_linear_fwd(0, N, N, N, 1, 0.2, N, 4, 64, 8, N, 0, st),
        lambda: lib.dsr_linear_dgrad(0, N, N, N, 4, 8, 64, st),
        lambda: lib.dsr_linear_wgrad(0, N, N, N, 32, 8, 64, st),
        lambda: lib.dsr_linear_wgrad_gathered(0, N, N, N, 32, 8, 64, 2, 0.5, st),
        lambda: lib.dsr_clock_sample(N, st),
        lambda: lib.dsr_conv_dgrad_masked(ctypes.byref(d), N, N, N, L.ACT_RELU, 0.0, N, st),
        lambda: lib.dsr_conv_dgrad_masked(ctypes.byref(d), one, one, one, L.ACT_LEAKY, 0.0, one, st),     # slope <= 0
        lambda: lib.dsr_conv_dgrad_masked(ctypes.byref(d), one, one, one, L.ACT_TANH, 0.0, one, st),      # not a sign-type activation
        lambda: lib.dsr_maxpool2_relu_bwd(0, N, N, N, 1, 4, 4, 8, st),
        lambda: lib.dsr_resample_u8(N, N, 4, 4, 3, 1, 2, N, N, 5, st),
        lambda: lib.dsr_resample_u8(one, one, 4, 4, 3, 2, 2, one, one, 5, st),                # axis
        lambda: lib.dsr_noise_gaussian_u8(N, N, 1, N, 16, st),
        lambda: lib.dsr_salt_pepper_u8(N, N, N, N, 4, 4, 3, st),
        lambda: lib.dsr_scale_images_f32(N, 16, 1, st),
        lambda: lib.dsr_scale_images_f32(one, 16, 0, st),                                      # mode
        lambda: lib.dsr_patch_batch_u8(1, None, None, None, None, None, 4, 4, 0, N, st),
        lambda: lib.dsr_patch_batch_u8(1, (ctypes.c_void_p * 1)(16), (ctypes.c_int * 1)(4), (ctypes.c_int * 1)(4), (ctypes.c_int * 1)(2),
                                       (ctypes.c_int * 1)(0), 4, 4, 0, one, st),               # rows 2..5 of a 4-row image
        lambda: lib.dsr_patch_batch_u8(1, (ctypes.c_void_p * 1)(16), (ctypes.c_int * 1)(4), (ctypes.c_int * 1)(4), (ctypes.c_int * 1)(0),
                                       (ctypes.c_int * 1)(0), 4, 4, 9, one, st),               # mode
        lambda: lib.dsr_conv_wgrad_batched(1, None, None, None, None, N, 0, st),
        lambda: lib.dsr_conv_wgrad_batched(0, ctypes.byref(d), None, None, None, N, 0, st),
        lambda: lib.dsr_conv_wgrad_batched(100000, ctypes.byref(d), None, None, None, N, 0, st),
        lambda: lib.dsr_linear_wgrad_adam(0, N, N, 32, 8, 64, 1, 1.0, N, N, N, N, N, 1e-3, 0.9, 0.999, 1e-8, 1.0, st),
        lambda: lib.dsr_linear_wgrad_adam(0, one, one, 32, 8, 40, 1, 1.0, one, one, one, N, one, 1e-3, 0.9, 0.999, 1e-8, 1.0, st),  # K % 64
        lambda: lib.dsr_dense2_fwd(N, N, N, 4, 8, N, st),
        lambda: lib.dsr_dense2_bwd(0, N, N, N, N, 4, 8, 32, 0.2, N, N, N, N, N, st),
        lambda: lib.dsr_maxpool2_fwd(0, N, N, 1, 4, 4, 8, st),
        lambda: lib.dsr_maxpool2_bwd(0, N, N, N, 1, 4, 4, 8, st),
        lambda: lib.dsr_avgpool2_fwd(0, N, N, 1, 4, 4, 8, st),
        lambda: lib.dsr_avgpool2_bwd(0, N, N, 1, 4, 4, 8, st),
        lambda: lib.dsr_nearest2x_fwd(0, N, N, 1, 4, 4, 8, st),
        lambda: lib.dsr_nearest2x_bwd(0, N, N, 1, 4, 4, 8, st),
        lambda: lib.dsr_bilinear2x_fwd(0, N, N, 1, 4, 4, 8, st),
        lambda: lib.dsr_bilinear2x_bwd(0, N, N, 1, 4, 4, 8, st),
        # mean3 / std3 are HOST arrays that the launcher reads: null used to be a host segfault
        lambda: lib.dsr_resize_norm_fwd(0, one, one, 1, 3, 8, 8, 4, 4, one, one, one, one, one, one, 2, N, N, st),
        lambda: lib.dsr_resize_norm_bwd(0, one, one, 1, 3, 8, 8, 4, 4, one, one, one, one, one, one, 2, N, st),
        lambda: lib.dsr_box_copy(N, N, 1, 2, 2, 4, 4, 4, 8, 0, 0, 0, 4, 4, 8, 0, 0, 0, st),
        lambda: lib.dsr_downsample_fwd(N, N, N, 3, 8, 8, 4, 2, 1, st),
        lambda: lib.dsr_downsample_bwd(N, N, N, 3, 8, 8, 4, 2, 1, st),
        lambda: lib.dsr_ssim_f32(N, N, 3, 32, 32, 1.0, N, st),
        lambda: lib.dsr_ssim_f32(one, one, 3, 8, 32, 1.0, one, st),           # image smaller than the 11x11 window
        lambda: lib.dsr_ssim_f32(one, one, 3, 32, 32, 0.0, one, st),          # data_range
    ]
    for i, call in enumerate(calls):
        rc = call()
        assert rc < 0, f"call #{i} returned {rc}"
        assert lib.dsr_last_error(), i
    # size queries on a bad descriptor answer 0 instead of dereferencing it
    assert lib.dsr_conv_packed_elems(N, 0) == 0 and lib.dsr_conv_dgrad_workspace(N) == 0
    assert lib.dsr_conv_wgrad_workspace(N) == 0 and lib.dsr_conv_first_bwd_workspace(N) == 0
    assert lib.dsr_conv_stats_rows(N) < 0 and lib.dsr_conv_fwd_affine_supported(N) == 0
    assert lib.dsr_conv_kernel_name(N, 0, N) == b"invalid"
    assert lib.dsr_ssim_blocks(3, 8, 32) == 0 and lib.dsr_ssim_blocks(3, 42, 18) == 3 * 4 * 1


def test_entry_point_without_return_does_not_compile(tmp_path):
    """The 06:58 round-1 segfault (functional.py:286, inside dsr_pw_bn_eval_affine): that entry point had lost its
    `return` statement -- undefined behaviour, hipcc -O3 emits no `ret` and the host runs off the function's end.  The
    build now carries -Werror=return-type; this checks the flag is there and does what it is there for."""
    b = importlib.import_module(PKG + "._build")
    assert "-Werror=return-type" in b.FLAGS
    src = tmp_path / "noret.hip"
    src.write_text('extern "C" int f(int x) { if (x) return 1; }\n')
    import subprocess
    r = subprocess.run(["hipcc"] + b.FLAGS + ["-c", str(src), "-o", str(tmp_path / "noret.o")], capture_output=True, text=True)
    assert r.returncode != 0 and "return" in r.stderr


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under the package may import it."""
    pk = os.path.join(ROOT, PKG)
    for dirpath, _, files in os.walk(pk):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), os.path.join(dirpath, f)


def test_ops_fail_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    gen = importlib.import_module(PKG + ".models.GAN.generator")
    g = gen.Generator(4, 1)
    with pytest.raises(RuntimeError):
        g(torch.zeros(1, 3, 8, 8))
