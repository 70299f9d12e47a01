"""ctypes binding of include/adcdiff.h.  Loading fails loudly: there is no fallback path."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ADCDIFF_LIB") or os.path.join(HERE, "libadcdiff.so")   # ADCDIFF_LIB: an instrumented build (measurement aid)

F32, BF16 = 0, 1
STREAM_XT, STREAM_STEP, STREAM_QNOISE, STREAM_TDRAW = 1, 2, 3, 4

c_fp = C.c_void_p   # device pointers travel as integers


class GnOut(C.Structure):
    _fields_ = [("gamma", c_fp), ("beta", c_fp), ("out", c_fp), ("ldo", C.c_int), ("group_size", C.c_int)]


class StepUpdate(C.Structure):
    _fields_ = [("kind", C.c_int), ("c", C.c_float * 5), ("x", c_fp), ("ldx", C.c_int), ("noise", c_fp), ("ldn", C.c_int),
                ("seed", C.c_uint64), ("step", C.c_uint32), ("row_offset", C.c_uint64)]


class GemmArgs(C.Structure):
    _fields_ = [("dtype", C.c_int), ("M", C.c_int), ("N", C.c_int), ("m_valid", C.c_int), ("nseg", C.c_int),
                ("A", c_fp * 3), ("lda", C.c_int * 3), ("K", C.c_int * 3), ("W", c_fp), ("ldw", C.c_int),
                ("bias", c_fp), ("rowbias", c_fp), ("ldrb", C.c_longlong), ("rowbias_idx", c_fp), ("rowbias_rows", C.c_int),
                ("residual", c_fp), ("ldr", C.c_int), ("act", C.c_int), ("out_f32", c_fp), ("ldo_f32", C.c_int),
                ("out_op", c_fp), ("ldo_op", C.c_int), ("n_gn", C.c_int), ("gn", GnOut * 2), ("gn_eps", C.c_float), ("tile_hint", C.c_int),
                ("split_k", C.c_int), ("split_stride", C.c_longlong), ("update", StepUpdate),
                ("splitk_scratch", c_fp), ("splitk_scratch_bytes", C.c_longlong)]


class DdpmCoef(C.Structure):
    _fields_ = [("sqrt_recip_alpha", C.c_float), ("beta", C.c_float), ("sqrt_one_minus_ac", C.c_float),
                ("sigma", C.c_float)]


class DdimCoef(C.Structure):
    _fields_ = [("c_eps", C.c_float), ("sqrt_ac_t", C.c_float), ("sqrt_ac_prev", C.c_float),
                ("dir_coef", C.c_float), ("sigma", C.c_float)]


BLOCK_FIELDS = ["norm1_w", "norm1_b", "norm2_w", "norm2_b", "linear1_w", "linear1_b", "linear2_w", "linear2_b",
                "time_proj_w", "time_proj_b", "residual_w", "residual_b"]


class BlockParams(C.Structure):
    _fields_ = [(f, c_fp) for f in BLOCK_FIELDS]


class UnetParams(C.Structure):
    _fields_ = [("input_dim", C.c_int), ("hidden_dim", C.c_int), ("output_dim", C.c_int), ("num_timesteps", C.c_int),
                ("time_lin1_w", c_fp), ("time_lin1_b", c_fp), ("time_lin2_w", c_fp), ("time_lin2_b", c_fp),
                ("cond0_w", c_fp), ("cond0_b", c_fp), ("cond2_w", c_fp), ("cond2_b", c_fp),
                ("blocks", BlockParams * 10),
                ("out_norm_w", c_fp), ("out_norm_b", c_fp), ("out_w", c_fp), ("out_b", c_fp),
                ("sinusoid_freq", c_fp)]


class BlockGrads(C.Structure):
    _fields_ = [(f, c_fp) for f in BLOCK_FIELDS]


class UnetGrads(C.Structure):
    _fields_ = [("time_lin1_w", c_fp), ("time_lin1_b", c_fp), ("time_lin2_w", c_fp), ("time_lin2_b", c_fp),
                ("blocks", BlockGrads * 10),
                ("out_norm_w", c_fp), ("out_norm_b", c_fp), ("out_w", c_fp), ("out_b", c_fp),
                ("cond0_w", c_fp), ("cond0_b", c_fp), ("cond2_w", c_fp), ("cond2_b", c_fp)]


class TimingRecord(C.Structure):
    _fields_ = [("kind", C.c_int), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("ms", C.c_float), ("flops", C.c_double),
                ("bytes", C.c_double)]


class OptTensor(C.Structure):
    _fields_ = [("param", c_fp), ("grad", c_fp), ("exp_avg", c_fp), ("exp_avg_sq", c_fp), ("numel", C.c_longlong),
                ("chunk_start", C.c_longlong)]


class AdamWArgs(C.Structure):
    _fields_ = [("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double), ("weight_decay", C.c_double),
                ("max_norm", C.c_double), ("step", C.c_int), ("decoupled", C.c_int)]


OPT_CHUNK = 4096
OPT_NOALLOC_MAX_CHUNKS = 65536
KIND_GEMM_F32, KIND_GEMM_BF16, KIND_GROUPNORM, KIND_UPDATE, KIND_GEMM_BF16_T256, KIND_OTHER, KIND_GEMM_BF16_W256 = 0, 1, 2, 3, 4, 5, 6

# name -> (restype, argtypes): every symbol include/adcdiff.h declares
SIGNATURES = {
    "adcdiff_abi_version": (C.c_int, []),
    "adcdiff_last_error": (C.c_char_p, []),
    "adcdiff_gemm": (C.c_int, [C.POINTER(GemmArgs), c_fp]),
    "adcdiff_gemm_tr": (C.c_int, [C.c_int, c_fp, C.c_longlong, c_fp, C.c_longlong, c_fp, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int,
                               C.c_longlong, c_fp]),
    "adcdiff_groupnorm_silu": (C.c_int, [c_fp, C.c_int, C.c_int, C.c_int, C.c_int, c_fp, c_fp, C.c_float, C.c_int, c_fp,
                                         C.c_int, c_fp, C.c_int, c_fp]),
    "adcdiff_philox_normal": (C.c_int, [c_fp, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, c_fp]),
    "adcdiff_ddpm_step": (C.c_int, [c_fp, C.c_int, c_fp, C.c_int, c_fp, C.c_int, c_fp, C.c_int, C.c_int, C.c_int, DdpmCoef,
                                    C.c_uint64, C.c_uint32, C.c_uint64, c_fp]),
    "adcdiff_ddim_step": (C.c_int, [c_fp, C.c_int, c_fp, C.c_int, c_fp, C.c_int, c_fp, C.c_int, C.c_int, C.c_int, DdimCoef,
                                    C.c_uint64, C.c_uint32, C.c_uint64, c_fp]),
    "adcdiff_q_sample": (C.c_int, [c_fp, C.c_int, c_fp, c_fp, c_fp, C.c_int, c_fp, C.c_int, c_fp, C.c_int, c_fp, C.c_int,
                                   C.c_int, C.c_int, C.c_uint64, C.c_uint32, C.c_uint64, c_fp]),
    "adcdiff_unet_arena_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "adcdiff_unet_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "adcdiff_unet_create": (C.c_int, [C.POINTER(UnetParams), C.c_int, c_fp, C.c_size_t, C.POINTER(c_fp)]),
    "adcdiff_unet_destroy": (None, [c_fp]),
    "adcdiff_unet_pack": (C.c_int, [c_fp, c_fp, C.c_size_t, c_fp]),
    "adcdiff_unet_condition_embed": (C.c_int, [c_fp, c_fp, C.c_int, C.c_int, c_fp, C.c_int, c_fp, C.c_size_t, c_fp]),
    "adcdiff_unet_forward": (C.c_int, [c_fp, c_fp, C.c_int, C.c_int, c_fp, C.c_int, c_fp, C.c_int, c_fp, C.c_int, c_fp,
                                       C.c_size_t, c_fp]),
    "adcdiff_ddpm_sample_loop": (C.c_int, [c_fp, c_fp, C.c_int, C.c_int, c_fp, C.c_int, C.POINTER(DdpmCoef), C.c_int, C.c_int,
                                           c_fp, C.c_longlong, C.c_uint64, C.c_uint64, c_fp, C.c_size_t, c_fp]),
    "adcdiff_ddim_sample_loop": (C.c_int, [c_fp, c_fp, C.c_int, C.c_int, c_fp, C.c_int, C.POINTER(C.c_int),
                                           C.POINTER(DdimCoef), C.c_int, c_fp, C.c_longlong, C.c_uint64, C.c_uint64, c_fp,
                                           C.c_size_t, c_fp]),
    "adcdiff_unet_train_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "adcdiff_unet_train_forward": (C.c_int, [c_fp, c_fp, C.c_int, C.c_int, c_fp, c_fp, C.c_int, C.c_float, C.c_uint64, c_fp, C.c_int, c_fp, C.c_size_t,
                                             c_fp]),
    "adcdiff_mse_loss_grad": (C.c_int, [c_fp, C.c_int, c_fp, C.c_int, c_fp, C.c_int, C.c_int, C.c_int, C.c_float, c_fp, c_fp]),
    "adcdiff_unet_backward": (C.c_int, [c_fp, c_fp, C.c_int, C.POINTER(UnetGrads), c_fp, C.c_size_t, c_fp]),
    "adcdiff_latent_summary": (C.c_int, [c_fp, C.c_int, C.c_int, C.c_int, c_fp, C.c_int, c_fp]),
    "adcdiff_clip_adamw_workspace_bytes": (C.c_size_t, [C.c_longlong]),
    "adcdiff_clip_adamw_step": (C.c_int, [c_fp, C.c_int, C.c_longlong, C.POINTER(AdamWArgs), c_fp, c_fp, C.c_size_t, c_fp]),
    "adcdiff_clip_adamw_step_noalloc": (C.c_int, [c_fp, C.c_int, C.c_longlong, C.POINTER(AdamWArgs), c_fp, c_fp]),
    "adcdiff_debug_set": (C.c_int, [C.c_char_p, C.c_longlong]),
    "adcdiff_debug_get": (C.c_int, [C.c_char_p, C.POINTER(C.c_longlong)]),
    "adcdiff_timing_begin": (C.c_int, [C.c_int]),
    "adcdiff_timing_end": (C.c_int, [C.POINTER(TimingRecord), C.c_int, C.POINTER(C.c_int)]),
    "adcdiff_matrix_core_probe": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}

_lib = None


class AdcdiffError(RuntimeError):
    pass


def load():
    """Load libadcdiff.so and bind every entry point; raises if the library was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AdcdiffError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
                           "There is no CPU or PyTorch fallback for this path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.adcdiff_abi_version() != 3:
        raise AdcdiffError("libadcdiff.so has ABI version %d, expected 3" % lib.adcdiff_abi_version())
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise AdcdiffError("adcdiff error %d: %s" % (rc, load().adcdiff_last_error().decode("utf-8", "replace")))


def debug_set(key, value):
    """A measurement / test switch of the library (adcdiff_debug_set); process-wide."""
    check(load().adcdiff_debug_set(key.encode(), int(value)))


def debug_get(key):
    v = C.c_longlong()
    check(load().adcdiff_debug_get(key.encode(), C.byref(v)))
    return int(v.value)


def matrix_core_probe(iters=20000, launches=40):
    """(TFLOP/s, shader clock in MHz) that the matrix cores of the current GPU sustain on register operands (adcdiff_matrix_core_probe)."""
    tf, mhz = C.c_double(), C.c_double()
    check(load().adcdiff_matrix_core_probe(int(iters), int(launches), C.byref(tf), C.byref(mhz)))
    return float(tf.value), float(mhz.value)
