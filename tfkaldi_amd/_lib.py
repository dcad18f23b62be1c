"""ctypes binding of the C-ABI engine (include/tfkaldi_hip.h).

There is deliberately NO fallback: if libtfkaldi_hip.so is missing or does not load, importing the
engine fails loudly -- the product path never routes through a CPU implementation.
"""
import ctypes
import os
from ctypes import (POINTER, Structure, byref, c_char, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_long,
                    c_size_t, c_uint64, c_void_p)

from .build import lib_path, source_id

ABI_VERSION = 8

# enums of tfkaldi_hip.h
NONLIN = {"relu": 0, "sigmoid": 1, "tanh": 2, "linear": 3}
# Arithmetic of the contractions (tfkaldi_hip.h: TFK_DTYPE_*; parameters, statistics, loss, gradient sums and Adam are fp32 in
# every mode):
#   "float32"       the reference's arithmetic (fp32 `tf.matmul`, layer.py:52).  Since round 5 it runs EMULATED on the bf16 matrix
#                   pipe -- every operand split exactly into three bf16 planes, six exact plane products accumulated in fp32
#                   (TFK_DTYPE_F32X3; error bound and evidence: DESIGN.md 4) -- which is closer to float64 than the fp32 matrix
#                   instruction chain and ~1.3x as fast.  "float32x3" names the same thing explicitly.  Two differences from the
#                   fp32 matrix instruction a caller should know: (1) NON-FINITE operands -- an Inf splits into (Inf, NaN, NaN), so
#                   an infinite activation or weight turns every result it touches into NaN where fp32 gives +-Inf (NaN stays
#                   NaN; finite operands whose products overflow give +-Inf as in fp32) -- a diverged run shows as NaN, not Inf;
#                   (2) MEMORY -- every GEMM operand (activations, weights, the loss gradient) has a three-plane twin of 6 bytes
#                   per element beside its 4: about +50 % of the activation memory and +6 B per weight (cfg2: ~160 MB of 288 GB).
#                   TFK_F32_ARITHMETIC=mfma (or "float32_mfma") has neither.
#   "float32_mfma"  the exact fp32 matrix instructions (v_mfma_f32_32x32x2_f32; TFK_DTYPE_F32): the default of rounds 1-4.
#   "bfloat16"      mixed precision: operands ROUNDED to bf16 (BASELINE cfg3 / cfg4).
# env TFK_F32_ARITHMETIC = mfma | x3 overrides what "float32" means for the whole process (a site's policy, A/B runs).
DTYPES = {"float32": 2, "float32x3": 2, "float32_mfma": 0, "bfloat16": 1}


def resolve_dtype(name):
    """the TFK_DTYPE_* value of a compute_dtype name"""
    if name not in DTYPES:
        raise ValueError("compute_dtype must be one of %s" % sorted(DTYPES))
    if name == "float32":
        policy = os.environ.get("TFK_F32_ARITHMETIC", "x3")
        if policy not in ("x3", "mfma"):
            raise ValueError("TFK_F32_ARITHMETIC=%r (x3 | mfma)" % policy)
        return 0 if policy == "mfma" else 2
    return DTYPES[name]


WEIGHTS, BIASES, BN_BETA, BN_MOVING_MEAN, BN_MOVING_VAR = range(5)
SLOT_PARAM, SLOT_GRAD, SLOT_ADAM_M, SLOT_ADAM_V = range(4)
(GLOBAL_STEP, LEARNING_RATE_FACT, INITIALISED_LAYERS, ADAM_STEPS, BATCH_LOSS, NUM_FRAMES,
 LEARNING_RATE) = range(7)
DEVICE_PTRS, LAST_MICROBATCH, LOG_DIV_PRIOR, RAW_LOGITS, RAW_DEVICE = 1, 2, 4, 8, 16
DBG_LOGITS, DBG_HIDDEN, DBG_DROPOUT_MASK, DBG_PREACT, DBG_BN_MEAN, DBG_BN_RSTD = range(6)
GEMM_NN, GEMM_NT, GEMM_TN = range(3)
EPI_BIAS, EPI_ACCUM, EPI_RELU = 1, 2, 4
EXCHANGE = {"sharded": 0, "allreduce": 1}  # TFK_EXCHANGE_*


class TfkConfig(Structure):
    _fields_ = [
        ("struct_size", c_int32), ("device", c_int32), ("input_dim", c_int32), ("num_layers", c_int32),
        ("num_units", c_int32), ("output_dim", c_int32), ("nonlin", c_int32), ("batch_norm", c_int32),
        ("l2_norm", c_int32), ("keep_prob", c_float), ("layerwise_init", c_int32),
        ("init_learning_rate", c_float), ("learning_rate_decay", c_float), ("num_steps", c_int32),
        ("max_frames", c_int32), ("seed", c_uint64), ("bn_decay", c_float), ("bn_epsilon", c_float),
        ("adam_beta1", c_float), ("adam_beta2", c_float), ("adam_epsilon", c_float),
        ("compute_dtype", c_int32),
    ]


class TfkKernelStat(Structure):
    _fields_ = [("name", c_char * 48), ("launches", c_int64), ("total_ms", c_double), ("flops", c_double),
                ("bytes", c_double)]


class TfkFeatConfig(Structure):
    """tfk_feat_config of include/tfkaldi_hip.h (feature computation plan)"""
    _fields_ = [
        ("struct_size", c_int32), ("device", c_int32), ("kind", c_int32), ("dynamic", c_int32),
        ("frame_len", c_int32), ("frame_step", c_int32), ("nfft", c_int32), ("nfilt", c_int32), ("numcep", c_int32),
        ("include_energy", c_int32), ("preemph", c_double),
    ]


FEAT_KIND = {"fbank": 0, "mfcc": 1, "ssc": 2, "fbank_raw": 3}   # TFK_FEAT_*
FEAT_DYNAMIC = {"nodelta": 0, "delta": 1, "ddelta": 2}          # TFK_DYN_*
SAMPLE_I16, SAMPLE_F64, SAMPLE_F32 = 0, 1, 2
STAGE_FRAMES, STAGE_MAGSPEC, STAGE_POWSPEC = 1, 2, 3

BUCKET_FN = ctypes.CFUNCTYPE(None, c_void_p, c_int)

# every symbol include/tfkaldi_hip.h declares: name -> (restype, argtypes)
_E = c_void_p
SYMBOLS = {
    "tfk_abi_version": (c_int, []),
    "tfk_last_error": (c_char_p, []),
    "tfk_build_id": (c_char_p, []),
    "tfk_state_bytes": (c_int, [POINTER(TfkConfig), POINTER(c_size_t)]),
    "tfk_create": (c_int, [POINTER(TfkConfig), POINTER(_E)]),
    "tfk_create_ex": (c_int, [POINTER(TfkConfig), c_void_p, c_size_t, c_void_p, POINTER(_E)]),
    "tfk_destroy": (c_int, [_E]),
    "tfk_tensor_count": (c_int, [_E, c_int, c_int, POINTER(c_size_t)]),
    "tfk_tensor_get": (c_int, [_E, c_int, c_int, c_int, c_void_p, c_size_t]),
    "tfk_tensor_set": (c_int, [_E, c_int, c_int, c_int, c_void_p, c_size_t]),
    "tfk_scalar_get": (c_int, [_E, c_int, POINTER(c_double)]),
    "tfk_scalar_set": (c_int, [_E, c_int, c_double]),
    "tfk_accumulate": (c_int, [_E, c_void_p, c_int64, c_void_p, c_int32, c_int]),
    "tfk_accumulate_raw": (c_int, [_E, c_void_p, c_int64, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_void_p,
                                   c_int]),
    "tfk_eval_accumulate_raw": (c_int, [_E, c_void_p, c_int64, c_void_p, c_int32, c_void_p, c_int32, c_int32,
                                        c_void_p, c_int]),
    "tfk_accumulate_stacked": (c_int, [_E, c_void_p, c_int64, c_void_p, c_int32, c_void_p, c_int32, c_int]),
    "tfk_accumulate_stacked_raw": (c_int, [_E, c_void_p, c_int64, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_void_p,
                                           c_void_p, c_int32, c_int]),
    "tfk_eval_accumulate_stacked": (c_int, [_E, c_void_p, c_int64, c_void_p, c_int32, c_void_p, c_int32, c_int]),
    "tfk_eval_accumulate_stacked_raw": (c_int, [_E, c_void_p, c_int64, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_void_p,
                                                c_void_p, c_int32, c_int]),
    "tfk_accumulate_ctc": (c_int, [_E, c_void_p, c_int64, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_int]),
    "tfk_eval_accumulate_ctc": (c_int, [_E, c_void_p, c_int64, c_int32, c_void_p, c_int32, c_void_p, c_void_p,
                                        c_int]),
    "tfk_accumulate_ctc_raw": (c_int, [_E, c_void_p, c_int64, c_int32, c_void_p, c_int32, c_int32, c_void_p, c_void_p,
                                       c_void_p, c_int]),
    "tfk_eval_accumulate_ctc_raw": (c_int, [_E, c_void_p, c_int64, c_int32, c_void_p, c_int32, c_int32, c_void_p,
                                            c_void_p, c_void_p, c_int]),
    "tfk_posteriors_raw": (c_int, [_E, c_void_p, c_int64, c_int32, c_void_p, c_int32, c_int32, c_void_p, c_void_p,
                                   c_int64, c_int]),
    "tfk_apply": (c_int, [_E, POINTER(c_float)]),
    "tfk_apply_enqueue": (c_int, [_E]),
    "tfk_eval_accumulate": (c_int, [_E, c_void_p, c_int64, c_void_p, c_int32, c_int]),
    "tfk_eval_finish": (c_int, [_E, POINTER(c_float)]),
    "tfk_halve_learning_rate": (c_int, [_E]),
    "tfk_add_layer": (c_int, [_E]),
    "tfk_init_last_layer": (c_int, [_E]),
    "tfk_posteriors": (c_int, [_E, c_void_p, c_int64, c_int32, c_void_p, c_int64, c_int]),
    "tfk_set_prior": (c_int, [_E, c_void_p, c_size_t]),
    "tfk_reduce_region": (c_int, [_E, POINTER(c_void_p), POINTER(c_size_t)]),
    "tfk_zero_accumulators": (c_int, [_E]),
    "tfk_reduce_bucket": (c_int, [_E, c_int, POINTER(c_size_t), POINTER(c_size_t)]),
    "tfk_num_buckets": (c_int, [_E, POINTER(c_int)]),
    "tfk_apply_begin": (c_int, [_E]),
    "tfk_apply_span": (c_int, [_E, c_size_t, c_size_t]),
    "tfk_apply_end": (c_int, [_E, POINTER(c_float)]),
    "tfk_set_bucket_callback": (c_int, [_E, BUCKET_FN, c_void_p]),
    "tfk_set_later_microbatches": (c_int, [_E, c_int32]),
    "tfk_params_touched": (c_int, [_E]),
    "tfk_twins_from_params": (c_int, [_E, c_size_t, c_size_t, c_void_p, POINTER(c_int)]),
    "tfk_set_layer_callback": (c_int, [_E, BUCKET_FN, c_void_p]),
    "tfk_shadow_region": (c_int, [_E, POINTER(c_void_p), POINTER(c_size_t), POINTER(c_int)]),
    "tfk_twin_region": (c_int, [_E, c_int, POINTER(c_void_p), POINTER(c_size_t), POINTER(c_int)]),
    "tfk_comm_set_gather": (c_int, [c_void_p, c_int]),
    "tfk_comm_set_bucket_bytes": (c_int, [c_void_p, c_size_t]),
    "tfk_apply_writes_shadow": (c_int, [_E, POINTER(c_int)]),
    "tfk_param_checksum": (c_int, [_E, c_int, POINTER(c_uint64)]),
    "tfk_param_region": (c_int, [_E, POINTER(c_void_p), POINTER(c_size_t)]),
    "tfk_moment_regions": (c_int, [_E, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_size_t)]),
    "tfk_comm_available": (c_int, [_E, c_int]),
    "tfk_comm_unique_id": (c_int, [c_void_p, c_size_t, POINTER(c_size_t)]),
    "tfk_comm_create": (c_int, [_E, c_void_p, c_size_t, c_int, c_int, c_int, c_size_t, POINTER(c_void_p)]),
    "tfk_comm_destroy": (c_int, [c_void_p]),
    "tfk_comm_info": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "tfk_comm_backend": (c_char_p, [c_void_p]),
    "tfk_comm_apply": (c_int, [c_void_p, POINTER(c_float)]),
    "tfk_comm_apply_enqueue": (c_int, [c_void_p]),
    "tfk_comm_apply_end": (c_int, [c_void_p, POINTER(c_float)]),
    "tfk_comm_eval_finish": (c_int, [c_void_p, POINTER(c_float)]),
    "tfk_comm_idle": (c_int, [c_void_p]),
    "tfk_comm_finish_reduce": (c_int, [c_void_p]),
    "tfk_comm_drain": (c_int, [c_void_p]),
    "tfk_comm_masters_stale": (c_int, [c_void_p, POINTER(c_int)]),
    "tfk_comm_gather_masters": (c_int, [c_void_p]),
    "tfk_comm_last_step": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_size_t), c_int,
                                   POINTER(c_int)]),
    "tfk_comm_set_exchange": (c_int, [c_void_p, c_int, c_int]),
    "tfk_comm_get_exchange": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_double)]),
    "tfk_comm_tune": (c_int, [c_void_p, c_size_t, c_int]),
    "tfk_comm_timing": (c_int, [c_void_p, c_int]),
    "tfk_comm_timing_read": (c_int, [c_void_p, POINTER(c_double), c_int, POINTER(c_long)]),
    "tfk_loopback_create": (c_int, [c_int, POINTER(c_void_p)]),
    "tfk_loopback_destroy": (c_int, [c_void_p]),
    "tfk_comm_create_loopback": (c_int, [_E, c_void_p, c_int, c_int, c_size_t, POINTER(c_void_p)]),
    "tfk_synchronize": (c_int, [_E]),
    "tfk_stream": (c_int, [_E, POINTER(c_void_p)]),
    "tfk_profile_begin": (c_int, [_E]),
    "tfk_profile_end": (c_int, [_E, POINTER(TfkKernelStat), c_int, POINTER(c_int)]),
    "tfk_debug_fetch": (c_int, [_E, c_int, c_int, c_void_p, c_size_t]),
    "tfk_gemm_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int,
                             c_int, c_void_p, c_int, c_int]),
    "tfk_gemm_bf16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int,
                              c_int, c_void_p, c_int]),
    "tfk_debug_poison_splitk": (c_int, [_E]),
    "tfk_split3": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int]),
    "tfk_gemm_bf16x3": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                c_int]),
    "tfk_gemm_bf16x3_dual": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                     c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int]),
    "tfk_gemm_bf16_dual": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                   c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int]),
    "tfk_gemm_bf16_dual_config": (c_int, [c_int, c_int, c_int, c_int]),
    "tfk_gemm_bf16_force_config": (c_int, [c_int]),
    "tfk_gemm_bf16_config": (c_int, [c_int, c_int]),
    "tfk_feat_create": (c_int, [POINTER(TfkFeatConfig), c_void_p, c_void_p, c_void_p, c_void_p, POINTER(_E)]),
    "tfk_feat_destroy": (c_int, [_E]),
    "tfk_feat_dim": (c_int, [_E, POINTER(c_int32)]),
    "tfk_feat_compute": (c_int, [_E, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int32, c_int64, c_void_p, c_int64,
                                 c_int]),
    "tfk_feat_stage": (c_int, [_E, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int32, c_int64, c_void_p,
                               c_int64]),
    "tfk_feat_dynamic": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_int32, c_int64, c_int, c_int, c_void_p,
                                 c_int64, c_int]),
    "tfk_deframesig": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int32, c_int32, c_void_p, c_void_p]),
    "tfk_logpow": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "tfk_cmvn_stats": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_void_p]),
}

_lib = None


class EngineError(RuntimeError):
    """A non-zero status from the C ABI (message from tfk_last_error())."""


def load():
    """dlopen libtfkaldi_hip.so, bind every declared symbol and check the ABI version."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # The engine shares device memory and streams with torch in one process, so both must bind the SAME
        # HIP runtime: torch's bundled libamdhip64 has to be loaded before this library is dlopen'ed.
        import torch  # noqa: F401
    except ImportError:
        pass
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(
            "%s not found: build the gfx950 engine first (python -m tfkaldi_amd.build, or "
            "__graft_entry__.build()); there is no CPU fallback" % path)
    lib = ctypes.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.tfk_abi_version() != ABI_VERSION:
        raise ImportError("libtfkaldi_hip.so ABI %d != binding ABI %d" % (lib.tfk_abi_version(), ABI_VERSION))
    if os.environ.get("TFK_ALLOW_STALE_LIB") != "1":
        built_from = lib.tfk_build_id().decode()
        try:
            tree = source_id()
        except (IOError, OSError) as exc:
            # a deployment that ships the built library without its sources (a wheel, a container layer): there is nothing to
            # compare the baked id with -- say so once and accept the library for what it says it is
            import warnings
            warnings.warn("tfkaldi_amd: cannot verify %s against its sources (%s); loading build %s as it is"
                          % (path, exc, built_from))
            tree = built_from
        if built_from != tree:
            raise ImportError("%s was built from other sources (its build id %s, this tree %s): rebuild it with python -m "
                              "tfkaldi_amd.build (TFK_ALLOW_STALE_LIB=1 loads it anyway)" % (path, built_from, tree))
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().tfk_last_error()
        raise EngineError("tfkaldi_hip status %d: %s" % (rc, msg.decode() if msg else "?"))


def make_config(input_dim, num_layers, num_units, output_dim, nonlin="relu", batch_norm=False, l2_norm=False,
                keep_prob=1.0, layerwise_init=False, init_learning_rate=1e-3, learning_rate_decay=1.0,
                num_steps=1, max_frames=1024, seed=0, device=0, compute_dtype="float32"):
    if nonlin not in NONLIN:
        raise Exception('unkown nonlinearity')  # spelling as neuralNetworks/nnet.py:65
    c = TfkConfig()
    c.struct_size = ctypes.sizeof(TfkConfig)
    c.device = device
    c.input_dim, c.num_layers, c.num_units, c.output_dim = input_dim, num_layers, num_units, output_dim
    c.nonlin = NONLIN[nonlin]
    c.batch_norm, c.l2_norm, c.layerwise_init = int(bool(batch_norm)), int(bool(l2_norm)), int(bool(layerwise_init))
    c.keep_prob = float(keep_prob)
    c.init_learning_rate, c.learning_rate_decay = float(init_learning_rate), float(learning_rate_decay)
    c.num_steps, c.max_frames, c.seed = int(num_steps), int(max_frames), int(seed)
    c.compute_dtype = resolve_dtype(compute_dtype)
    return c
