"""ctypes binding of libbt_hip.so (include/bt_hip.h).  The HIP extension is mandatory: there is no CPU or PyTorch
fallback for the decode arithmetic — a missing library or a non-GPU tensor raises."""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_uint32, c_uint64, c_void_p

import torch  # noqa: F401  (must be imported first: libbt_hip.so binds to the HIP runtime torch already loaded)

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libbt_hip.so")
# the same sources compiled with -DBT_DEBUG_HOOKS: bt_debug_* setters + A/B / ablation kernels; tests and probes only
DEBUG_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libbt_hip_dbg.so")

BT_F32, BT_BF16 = 0, 1
BT_DEV_ERR_CACHE_FULL, BT_DEV_ERR_BLOCK_RANGE, BT_DEV_ERR_HANDOFF_TIMEOUT = 1, 2, 4
BT_ACT_NONE, BT_ACT_GELU = 0, 1
ATTN_ROWS, ATTN_FLASH = 0, 1


class BtError(RuntimeError):
    pass


class bt_decoder_dims(Structure):
    _fields_ = [("hidden", c_int32), ("heads", c_int32), ("head_dim", c_int32), ("rot", c_int32),
                ("inter", c_int32), ("layers", c_int32), ("ln_eps", c_float)]


class bt_layer_weights(Structure):
    _fields_ = [(n, c_void_p) for n in ("ln1_w", "ln1_b", "ln2_w", "ln2_b", "qkv_w", "qkv_b", "dense_w", "dense_b",
                                        "h4_w", "h4_b", "h_w", "h_b",
                                        "qkv_wf", "qkv_s", "qkv_c", "h4_wf", "h4_s", "h4_c")]   # folded LayerNorm (optional)


class bt_decoder_weights(Structure):
    _fields_ = [("layers_host", POINTER(bt_layer_weights)), ("lnf_w", c_void_p), ("lnf_b", c_void_p)]


class bt_engine_desc(Structure):
    """include/bt_hip.h bt_engine_desc (field order and types must match the header exactly)."""
    _fields_ = [("struct_size", c_uint32), ("dtype", c_int32), ("L", c_int32), ("E", c_int32), ("vocab", c_int32), ("B", c_int64),
                ("bd_dims", bt_decoder_dims), ("bd_w", bt_decoder_weights),
                ("td_dims", bt_decoder_dims), ("td_w", bt_decoder_weights),
                ("emb_table", c_void_p), ("emb_width", c_int32), ("do_sample", c_int32),
                ("exp_w", c_void_p), ("exp_b", c_void_p), ("embed_in", c_void_p), ("embed_out", c_void_p),
                ("gk", c_void_p), ("gv", c_void_p), ("g_layer_stride", c_int64), ("g_nmax", c_int64),
                ("g_cos", c_void_p), ("g_sin", c_void_p),
                ("lk", c_void_p), ("lv", c_void_p), ("l_layer_stride", c_int64), ("l_nmax", c_int64),
                ("l_cos", c_void_p), ("l_sin", c_void_p), ("l_pos_table", c_void_p),
                ("x_blk", c_void_p), ("ctx", c_void_p), ("x_pre", c_void_p), ("x_tok", c_void_p), ("hid", c_void_p),
                ("logits", c_void_p), ("raw", c_void_p), ("out_ids", c_void_p), ("nb_max", c_int64),
                ("n_cur", c_void_p), ("finished", c_void_p), ("finished_prev", c_void_p), ("mask", c_void_p),
                ("mask_ld", c_int64),
                ("ws_tok", c_void_p), ("ws_tok_bytes", c_int64), ("ws_step", c_void_p), ("ws_step_bytes", c_int64),
                ("exp_scratch", c_void_p), ("exp_scratch_bytes", c_int64),
                ("n_split", c_int32), ("top_k", c_int32), ("split_ws", c_void_p), ("split_ws_bytes", c_int64),
                ("temperature", c_float), ("skip_logits", c_int32), ("sample_row0", c_int64), ("seed", c_uint64), ("sample_step", c_void_p),
                ("config_pad", c_int64), ("gen_eos", c_int64), ("gen_pad", c_int64),
                ("row_map", c_void_p), ("aux_stream", c_void_p), ("ev_fork", c_void_p), ("ev_join", c_void_p), ("err_flag", c_void_p)]

    def __init__(self, *args, **kw):
        super().__init__(ctypes.sizeof(type(self)), *args, **kw)    # struct_size: the library rejects any other value


class bt_attn_args(Structure):
    _fields_ = [("struct_size", c_uint32),
                ("k_cache", c_void_p), ("v_cache", c_void_p), ("layer_stride", c_int64), ("n_max", c_int64),
                ("pos0_dev", c_void_p), ("pos0_host", c_int32), ("pos_table", c_void_p), ("mask", c_void_p),
                ("mask_ld", c_int64), ("cos_tab", c_void_p), ("sin_tab", c_void_p), ("mode", c_int32),
                ("n_split", c_int32), ("split_ws", c_void_p), ("split_ws_bytes", c_int64), ("seq_skip", c_void_p),
                ("aux_stream", c_void_p), ("ev_fork", c_void_p), ("ev_join", c_void_p), ("err_flag", c_void_p)]

    def __init__(self, *args, **kw):
        super().__init__(ctypes.sizeof(type(self)), *args, **kw)    # struct_size first; positional fields follow as before


_SIGS = {
    "bt_version": (c_int, []),
    "bt_last_error": (c_char_p, []),
    "bt_decoder_workspace_bytes": (c_int64, [POINTER(bt_decoder_dims), c_int, c_int64]),
    "bt_linear": (c_int, [c_int, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64,
                          c_void_p, c_int64, c_int, c_void_p]),
    "bt_linear_ws": (c_int, [c_int, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64,
                             c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p]),
    "bt_layernorm": (c_int, [c_int, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64,
                             c_void_p, c_void_p, c_void_p, c_int64, c_float, c_void_p]),
    "bt_embed_rows": (c_int, [c_int, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64,
                              c_void_p, c_int32, c_void_p]),
    "bt_rope_kv_append": (c_int, [c_int, c_int64, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int64, c_void_p,
                                  c_void_p, c_int64, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "bt_qkv_rope_append": (c_int, [c_int, c_int64, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int64, c_void_p, c_void_p,
                                   c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int32, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_int64, c_void_p]),
    "bt_ln_linear": (c_int, [c_int, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_float, c_void_p,
                             c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p]),
    "bt_linear2": (c_int, [c_int, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                           c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    "bt_stream_k_scratch_bytes": (c_int64, []),
    "bt_stream_k_reset": (c_int, [c_void_p, c_int64, c_void_p]),
    "bt_stream_k_status": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "bt_linear2_ws": (c_int, [c_int, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                              c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    "bt_ln_qkv_rope_append": (c_int, [c_int, c_int64, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int64, c_void_p,
                                      c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64,
                                      c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "bt_attention": (c_int, [c_int, c_int, c_int64, c_int32, c_int32, c_int32, c_void_p, c_int64, c_void_p, c_void_p,
                             c_int64, c_void_p, c_int32, c_void_p, c_int64, c_void_p, c_int64, c_int32, c_void_p,
                             c_int64, c_void_p]),
    "bt_attention_split_count": (c_int32, [c_int32, c_int64, c_int32]),
    "bt_argmax": (c_int, [c_int, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    "bt_logits_argmax_workspace_bytes": (c_int64, [c_int64, c_int64]),
    "bt_logits_argmax": (c_int, [c_int, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int32,
                                 c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    "bt_topk_sample": (c_int, [c_int, c_int64, c_int64, c_void_p, c_int64, c_float, c_int32, c_uint64, c_void_p, c_int64,
                               c_void_p, c_int64, c_void_p]),
    "bt_decoder_forward": (c_int, [c_int, POINTER(bt_decoder_dims), POINTER(bt_decoder_weights), c_int64, c_int32,
                                   c_void_p, POINTER(bt_attn_args), c_int32, c_void_p, c_void_p, c_int64, c_void_p]),
    "bt_finish_block": (c_int, [c_int64, c_int32, c_int32, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p]),
    "bt_bump": (c_int, [c_void_p, c_int32, c_void_p]),
    "bt_prefill_enqueue": (c_int, [POINTER(bt_engine_desc), c_void_p, c_int32, c_int64, c_int64, c_void_p, c_void_p, c_int64,
                                   c_void_p]),
    "bt_token_block_enqueue": (c_int, [POINTER(bt_engine_desc), c_int32, c_int32, c_int32, c_void_p]),
    "bt_roll_over_enqueue": (c_int, [POINTER(bt_engine_desc), c_void_p]),
    "bt_block_step_enqueue": (c_int, [POINTER(bt_engine_desc), c_int32, c_int32, c_void_p]),
    "bt_compact_enqueue": (c_int, [POINTER(bt_engine_desc), POINTER(bt_engine_desc), c_void_p, c_int64, c_void_p]),
    "bt_graph_begin_capture": (c_int, [c_void_p]),
    "bt_graph_end_capture": (c_int, [c_void_p, POINTER(c_void_p)]),
    "bt_graph_launch": (c_int, [c_void_p, c_void_p]),
    "bt_graph_destroy": (c_int, [c_void_p]),
    "bt_event_create": (c_int, [POINTER(c_void_p)]),
    "bt_event_record": (c_int, [c_void_p, c_void_p]),
    "bt_event_elapsed_ms": (c_int, [c_void_p, c_void_p, POINTER(c_float)]),
    "bt_event_destroy": (c_int, [c_void_p]),
    "bt_ceiling_copy": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_void_p]),
    "bt_ceiling_mfma_flops": (c_int64, [c_int32, c_int32]),
    "bt_ceiling_mfma": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32, c_void_p]),
    "bt_prof_enable": (c_int, [c_int]),
    "bt_prof_collect": (c_int, [POINTER(ctypes.c_double), POINTER(c_int64), c_int]),
    "bt_prof_trace": (c_int, [POINTER(ctypes.c_double), POINTER(ctypes.c_double), POINTER(ctypes.c_int32), c_int64,
                            POINTER(c_int64)]),
}

# test / A-B hooks: exported by libbt_hip_dbg.so only
_DEBUG_SIGS = {
    "bt_debug_force_generic_gemm": (None, [c_int]),
    "bt_debug_gemm_variant": (None, [c_int]),
    "bt_debug_gemm_max_wg": (None, [c_int]),
    "bt_debug_skinny_rows": (None, [c_int]),
    "bt_debug_small_stages": (None, [c_int]),
    "bt_debug_mid_rows": (None, [c_int]),
    "bt_debug_few_dual": (None, [c_int]),
    "bt_debug_few_persist": (None, [c_int]),
    "bt_debug_few_attn": (None, [c_int]),
    "bt_debug_mid_fuse": (None, [c_int]),
    "bt_debug_mid_dual": (None, [c_int]),
    "bt_debug_mid_attn": (None, [c_int]),
    "bt_debug_mid_tile": (None, [c_int]),
    "bt_debug_mid_force2": (None, [c_int]),
    "bt_debug_mid_tall": (None, [c_int]),
    "bt_debug_mid_tall_wgs": (None, [c_int]),
    "bt_debug_mid_tall_wt": (None, [c_int]),
    "bt_debug_mid_tall_fuse": (None, [c_int]),
    "bt_debug_mid_wt": (None, [c_int]),
    "bt_debug_mid_rows_max": (None, [c_int]),
    "bt_debug_mid_tall_wgs_res": (None, [c_int]),
    "bt_debug_no_qkv_fusion": (None, [c_int]),
    "bt_debug_flash_variant": (None, [c_int]),
    "bt_debug_attn_nt": (None, [c_int]),
    "bt_debug_attn_persist": (None, [c_int]),
    "bt_debug_local_kernel": (None, [c_int]),
    "bt_debug_attn_wide": (None, [c_int]),
    "bt_debug_few_rows_max": (None, [c_int]),
    "bt_debug_no_few_rows_fusion": (None, [c_int]),
    "bt_debug_lin2_large": (None, [c_int]),
    "bt_debug_lin2_sk": (None, [c_int]),
    "bt_debug_ln_fold": (None, [c_int]),
    "bt_debug_rowstats": (c_int, [c_int, c_int64, c_int64, c_void_p, c_int64, c_float, c_void_p, c_void_p]),
    "bt_debug_lnfold_shape_ok": (c_int32, [c_int64, c_int64, c_int64]),
    "bt_debug_linear_lnfold": (c_int, [c_int, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_int64, c_int, c_void_p]),
    "bt_debug_stream_k": (None, [c_int]),
    "bt_debug_stream_k_drop": (None, [c_int, c_int]),
    "bt_debug_timeline_buffer": (None, [c_void_p]),
    "bt_debug_serial_epilogues": (None, [c_int]),
    "bt_debug_sk_diag_buffer": (None, [c_void_p]),
    "bt_debug_no_two_branch": (None, [c_int]),
    "bt_debug_ln_streaming": (None, [c_int]),
    "bt_debug_tr_probe": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "bt_debug_grid_barrier_probe": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "bt_debug_memset": (c_int, [c_void_p, c_int32, c_int64, c_void_p]),
    "bt_debug_phase_probe": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p]),
}

# every symbol include/bt_hip.h declares (tests check the library exports all of them)
EXPORTED_SYMBOLS = list(_SIGS)

_handles = {}
_debug_active = False


def _load(path: str, debug: bool) -> ctypes.CDLL:
    if path not in _handles:
        if not os.path.exists(path):
            raise ImportError(
                f"{path} is missing: the Block Transformer decode path has no fallback without its HIP kernels. "
                "Build it with `python -m block_transformer_amd.build` (needs hipcc, no GPU required).")
        handle = ctypes.CDLL(path)
        sigs = dict(_SIGS, **_DEBUG_SIGS) if debug else _SIGS
        for name, (res, args) in sigs.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _handles[path] = handle
    return _handles[path]


def lib() -> ctypes.CDLL:
    """Load (once) and return the active library — the product libbt_hip.so unless a test switched to the debug build;
    raises ImportError with the build recipe when it is missing."""
    return _load(DEBUG_LIB_PATH, True) if _debug_active else _load(LIB_PATH, False)


def use_debug_library(on: bool = True) -> bool:
    """tests / tools only: route every op through libbt_hip_dbg.so (same kernels + bt_debug_* hooks).  Returns the
    previous setting.  Engines that captured hipGraphs keep replaying the kernels they were captured with."""
    global _debug_active
    prev, _debug_active = _debug_active, bool(on)
    if on:
        _load(DEBUG_LIB_PATH, True)
    return prev


PROF_FAMILIES = ["linear_mfma", "linear_generic", "layernorm", "embed", "rope_kv_append", "attn_rows", "attn_flash",
                 "argmax", "sample", "finish_block"]


def prof_enable(on: bool) -> None:
    check(lib().bt_prof_enable(1 if on else 0), "bt_prof_enable")


def prof_collect() -> dict:
    """{family: (milliseconds, launches)} accumulated since prof_enable(True)."""
    n = len(PROF_FAMILIES)
    ms = (ctypes.c_double * n)()
    cnt = (c_int64 * n)()
    check(lib().bt_prof_collect(ms, cnt, n), "bt_prof_collect")
    return {PROF_FAMILIES[i]: (ms[i], cnt[i]) for i in range(n)}


def prof_trace() -> list:
    """[(family, start_ms, dur_ms)] of every bracketed launch since prof_enable(True), in launch order (call before prof_collect)."""
    n = c_int64(0)
    check(lib().bt_prof_trace(None, None, None, 0, ctypes.byref(n)), "bt_prof_trace")
    if n.value == 0:
        return []
    start = (ctypes.c_double * n.value)()
    dur = (ctypes.c_double * n.value)()
    fam = (ctypes.c_int32 * n.value)()
    check(lib().bt_prof_trace(start, dur, fam, n.value, ctypes.byref(n)), "bt_prof_trace")
    return [(PROF_FAMILIES[fam[i]], start[i], dur[i]) for i in range(n.value)]


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().bt_last_error()
        raise BtError(f"libbt_hip {what} failed (status {rc}): {msg.decode() if msg else ''}")


def dtype_code(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return BT_F32
    if dtype == torch.bfloat16:
        return BT_BF16
    raise BtError(f"dtype {dtype} is not supported by libbt_hip (float32 and bfloat16 only)")


def ptr(t) -> int:
    """Device pointer of a tensor (None -> NULL).  Refuses CPU tensors: there is no CPU path."""
    if t is None:
        return None
    if not t.is_cuda:
        raise BtError("libbt_hip operates on GPU memory only; got a CPU tensor (move the model/inputs to 'cuda')")
    return t.data_ptr()


def current_stream() -> int:
    return torch.cuda.current_stream().cuda_stream
