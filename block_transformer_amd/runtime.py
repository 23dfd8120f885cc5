"""Host-side runtime pieces shared by the module API and the generate() engine: KV caches, the per-stack weight
tables handed to ``bt_decoder_forward``, activation workspaces and the hipGraph wrapper.

All device memory comes from torch's caching allocator (so ``torch.cuda.max_memory_allocated`` and OOM exceptions
behave like the reference's benchmark expects, measure_generation_time.py:205,237-249); the library never allocates.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional

import torch

from . import _lib
from ._lib import (bt_attn_args, bt_decoder_dims, bt_decoder_weights, bt_layer_weights, check, current_stream,
                   dtype_code, lib, ptr)
from .ops import rope_tables


class KVCache:
    """Preallocated cache K,V[layer][B][heads][n_max][d].  ``length`` = rows already filled (host-side bookkeeping of
    the module-level API; the generate() engine keeps its count in a device scalar instead)."""

    def __init__(self, layers: int, batch: int, heads: int, n_max: int, head_dim: int, dtype, device):
        self.k = torch.empty(layers, batch, heads, n_max, head_dim, dtype=dtype, device=device)
        self.v = torch.empty_like(self.k)
        self.length = 0

    @property
    def n_max(self) -> int:
        return self.k.shape[3]

    @property
    def batch(self) -> int:
        return self.k.shape[1]

    @property
    def layer_stride(self) -> int:
        return self.k.stride(0)

    @classmethod
    def rows_of(cls, parent: "KVCache", batch: int) -> "KVCache":
        """Sequences [0, batch) of ``parent``, SHARING its storage (views: the layer stride stays the parent's).  Used when the
        decode engine compacts its batch: the surviving sequences are moved to the front of the same cache."""
        assert 0 < batch <= parent.batch
        self = cls.__new__(cls)
        self.k = parent.k[:, :batch]
        self.v = parent.v[:, :batch]
        self.length = parent.length
        return self

    def grown(self, n_max: int) -> "KVCache":
        """A larger cache holding the same rows (only the module-level API ever grows a cache)."""
        L, B, Hh, _, d = self.k.shape
        new = KVCache(L, B, Hh, n_max, d, self.k.dtype, self.k.device)
        new.k[:, :, :, :self.length].copy_(self.k[:, :, :, :self.length])
        new.v[:, :, :, :self.length].copy_(self.v[:, :, :, :self.length])
        new.length = self.length
        return new

    # tuple-style access so callers written against HF tuple caches (past[0][0].shape[2]) keep working
    def __getitem__(self, layer: int):
        return (self.k[layer][:, :, :self.length], self.v[layer][:, :, :self.length])

    def __len__(self):
        return self.k.shape[0]


class Workspace:
    """A grow-only scratch tensor per device (activation workspace of bt_decoder_forward)."""

    def __init__(self):
        self._buf: Optional[torch.Tensor] = None

    def get(self, nbytes: int, device) -> torch.Tensor:
        if self._buf is None or self._buf.numel() < nbytes or self._buf.device != torch.device(device):
            self._buf = None
            self._buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        return self._buf


_shared_ws = Workspace()


def _version(p: torch.Tensor) -> int:
    """In-place modification counter of a parameter (inference tensors do not keep one: 0)."""
    try:
        return int(p._version)
    except RuntimeError:
        return 0


class DecoderRuntime:
    """ctypes view of one GPT-NeoX stack's parameters (HF layout, used in place — no repack, no copy)."""

    def __init__(self, gpt_neox: torch.nn.Module, config):
        self.module = gpt_neox
        self.config = config
        heads = config.num_attention_heads
        self.dims = bt_decoder_dims(config.hidden_size, heads, config.hidden_size // heads,
                                    int((config.hidden_size // heads) * config.rotary_pct), config.intermediate_size,
                                    config.num_hidden_layers, float(config.layer_norm_eps))
        self._key = None
        self._layers = None
        self._weights = None
        self._tables = {}
        self._err = {}            # device -> int32[1]: sticky BT_DEV_ERR_* bits of the passes run through forward() (check_errors)
        self._err_armed = False
        # per layer (qkv_wf, qkv_s, qkv_c, h4_wf, h4_s, h4_c): LayerNorm folded into its consumers -- a round-5 experiment that measured
        # no net gain (profiles/r5_ab_ln_fused.jsonl): built only on request AND with the debug library active (tests, A/B); the
        # product library ignores the fields
        self._folded = None
        self.fold_layernorm = False

    def _params(self) -> List[torch.Tensor]:
        out = []
        for layer in self.module.layers:
            out += [layer.input_layernorm.weight, layer.input_layernorm.bias,
                    layer.post_attention_layernorm.weight, layer.post_attention_layernorm.bias,
                    layer.attention.query_key_value.weight, layer.attention.query_key_value.bias,
                    layer.attention.dense.weight, layer.attention.dense.bias,
                    layer.mlp.dense_h_to_4h.weight, layer.mlp.dense_h_to_4h.bias,
                    layer.mlp.dense_4h_to_h.weight, layer.mlp.dense_4h_to_h.bias]
        out += [self.module.final_layer_norm.weight, self.module.final_layer_norm.bias]
        return out

    def weight_key(self):
        """Identity of the parameter storage (device pointers + dtype): engines key their C descriptor and their captured
        hipGraphs on it, so re-materialised parameters (``.to()``, ``load_state_dict(assign=True)``) are never replayed
        through stale pointers."""
        params = self._params()
        fold = self.fold_layernorm and _lib._debug_active and params[0].dtype == torch.bfloat16 and params[0].is_cuda
        return tuple(p.data_ptr() for p in params) + (params[0].dtype, fold) + (tuple(_version(p) for p in params) if fold else ())

    def _fold(self, params):
        """LayerNorm folded into the two projections that consume it (bt_layer_weights.*_wf / *_s / *_c, include/bt_hip.h): for a
        bf16 stack, per layer  W' = W diag(gamma) (bf16),  s = row sums of W' (of the ROUNDED values: what the GEMM multiplies),
        c = W beta + b  (fp32).  One-time weight preparation with torch ops -- not on the hot path; the HF-layout parameters stay
        untouched (the fold costs 7 H^2 extra bf16 elements per layer: 1.4 GB for both 1.2B stacks).  HF GPTNeoXLayer:
        input_layernorm -> attention.query_key_value, post_attention_layernorm -> mlp.dense_h_to_4h."""
        out = []
        with torch.no_grad():
            for i in range(len(self.module.layers)):
                ln1w, ln1b, ln2w, ln2b, qw, qb, _, _, hw, hb, _, _ = params[12 * i:12 * i + 12]
                rec = []
                for w, b, g, beta in ((qw, qb, ln1w, ln1b), (hw, hb, ln2w, ln2b)):
                    wf = (w.float() * g.float()[None, :]).to(w.dtype).contiguous()
                    svec = wf.float().sum(dim=1).contiguous()
                    cvec = (w.float() @ beta.float() + b.float()).contiguous()
                    rec += [wf, svec, cvec]
                out.append(rec)
        return out

    def weights(self) -> bt_decoder_weights:
        """(Re)build the pointer table when parameters moved (``.to(device)``, dtype cast, checkpoint load) -- and, because the folded
        LayerNorm weights are COPIES, also when a parameter was modified in place (its version counter moved)."""
        params = self._params()
        fold = self.fold_layernorm and _lib._debug_active and params[0].dtype == torch.bfloat16 and params[0].is_cuda
        key = tuple(p.data_ptr() for p in params) + (params[0].dtype, fold) + (tuple(_version(p) for p in params) if fold else ())
        if key != self._key:
            dt = params[0].dtype
            for p in params:
                if not p.is_cuda:
                    raise _lib.BtError("Block Transformer decoders run on the GPU only: call model.to('cuda') first")
                if p.dtype != dt or not p.is_contiguous():
                    raise _lib.BtError("all decoder parameters must share one dtype and be contiguous")
            n = len(self.module.layers)
            arr = (bt_layer_weights * n)()
            self._folded = self._fold(params) if fold else None
            for i in range(n):
                extra = [t.data_ptr() for t in self._folded[i]] if fold else []
                arr[i] = bt_layer_weights(*([p.data_ptr() for p in params[12 * i:12 * i + 12]] + extra))
            self._layers = arr
            self._weights = bt_decoder_weights(ctypes.cast(arr, ctypes.POINTER(bt_layer_weights)),
                                               params[-2].data_ptr(), params[-1].data_ptr())
            self._key = key
        return self._weights

    @property
    def dtype(self) -> torch.dtype:
        return self.module.final_layer_norm.weight.dtype

    @property
    def device(self):
        return self.module.final_layer_norm.weight.device

    def rope(self, max_pos: int) -> torch.Tensor:
        """[2, max_pos, rot/2] fp32 table on the parameters' device (cached, grown geometrically)."""
        dev = self.device
        t = self._tables.get(dev)
        if t is None or t.shape[1] < max_pos:
            cap = max(64, max_pos if t is None else max(max_pos, 2 * t.shape[1]))
            t = rope_tables(cap, self.dims.rot, float(getattr(self.config, "rotary_emb_base", 10000.0)), dev)
            self._tables[dev] = t
        return t

    def workspace_bytes(self, rows: int) -> int:
        return int(lib().bt_decoder_workspace_bytes(ctypes.byref(self.dims), dtype_code(self.dtype), rows))

    def forward(self, x: torch.Tensor, q_len: int, cache: KVCache, b0: int = 0, pos0: int = 0,
                pos0_dev: Optional[torch.Tensor] = None, pos_table: Optional[torch.Tensor] = None,
                mask: Optional[torch.Tensor] = None, mode: int = _lib.ATTN_ROWS, n_split: int = 1,
                split_ws: Optional[torch.Tensor] = None, first_out_row: int = 0, out: Optional[torch.Tensor] = None,
                ws: Optional[torch.Tensor] = None, rope: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Run the stack over x[B*q_len, H] (overwritten).  Sequences b0..b0+B of ``cache`` are read/extended at rows
        pos0..pos0+q_len; ``mask`` is uint8 [B, >= pos0+q_len] (already sliced to these B sequences)."""
        H = self.dims.hidden
        M = x.numel() // H
        B = M // q_len
        if x.dtype != self.dtype or not x.is_contiguous():
            raise _lib.BtError(f"activations must be contiguous {self.dtype}")
        if pos0_dev is None and pos0 + q_len > cache.n_max:
            raise _lib.BtError(f"KV cache overflow: {pos0}+{q_len} > {cache.n_max}")
        if ws is None:
            ws = _shared_ws.get(self.workspace_bytes(M), x.device)
        if rope is None:
            rope = self.rope(cache.n_max if pos_table is None else int(cache.n_max) + 64)
        count = q_len - first_out_row
        if out is None:
            out = torch.empty(B * count, H, dtype=x.dtype, device=x.device)
        es = x.element_size()
        seq_off = b0 * cache.k.stride(1) * es
        args = bt_attn_args(cache.k.data_ptr() + seq_off, cache.v.data_ptr() + seq_off, cache.layer_stride,
                            cache.n_max, ptr(pos0_dev), int(pos0), ptr(pos_table), ptr(mask),
                            0 if mask is None else mask.stride(0), rope[0].data_ptr(), rope[1].data_ptr(), int(mode),
                            int(n_split), ptr(split_ws), 0 if split_ws is None else split_ws.numel() * split_ws.element_size())
        if M > 256 and x.dtype == torch.bfloat16:   # the pass may run stream-K GEMMs: give their hand-off a place to report a time-out
            err = self._err.get(x.device)
            if err is None:
                err = self._err[x.device] = torch.zeros(1, dtype=torch.int32, device=x.device)
            args.err_flag = err.data_ptr()
            self._err_armed = True
        w = self.weights()
        check(lib().bt_decoder_forward(dtype_code(x.dtype), ctypes.byref(self.dims), ctypes.byref(w), B, q_len, ptr(x),
                                       ctypes.byref(args), first_out_row, ptr(out), ptr(ws), ws.numel(),
                                       current_stream()), "bt_decoder_forward")
        return out


    def check_errors(self) -> None:
        """Raise if a pass since the last check set a sticky device-side error bit (bt_attn_args.err_flag): today a stream-K GEMM
        hand-off time-out, after which the activations of that pass are incomplete.  Synchronises; the module-level ``forward``s
        call it once per call, only when a pass that could set the bit ran."""
        if not self._err_armed:
            return
        self._err_armed = False
        for err in self._err.values():
            bits = int(err.item())
            if bits:
                err.zero_()
                raise _lib.BtError(f"device-side guard tripped (bits {bits}): a stream-K GEMM hand-off timed out -- the GPU is "
                                   "shared with another process or a profiler? -- the outputs of this call are incomplete")


class Graph:
    """A captured hipGraph (``bt_graph_*``): capture once on torch's current stream, replay per block step."""

    def __init__(self):
        self._g = ctypes.c_void_p()
        self._stream = None

    def __enter__(self):
        self._stream = current_stream()
        check(lib().bt_graph_begin_capture(self._stream), "bt_graph_begin_capture")
        return self

    def __exit__(self, et, ev, tb):
        rc = lib().bt_graph_end_capture(self._stream, ctypes.byref(self._g))
        if et is None:
            check(rc, "bt_graph_end_capture")
        return False

    def launch(self):
        check(lib().bt_graph_launch(self._g, current_stream()), "bt_graph_launch")

    def __del__(self):
        try:
            if self._g:
                lib().bt_graph_destroy(self._g)
        except Exception:
            pass
