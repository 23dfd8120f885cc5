"""Tensor-level wrappers over the C-ABI single ops (one call = one or two HIP kernels on torch's current stream).

torch is used for device memory and streams only; no arithmetic runs in ATen here."""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from ._lib import check, current_stream, dtype_code, lib, ptr


def _rows(t: torch.Tensor):
    if t.dim() < 2 or t.stride(-1) != 1:
        raise _lib.BtError("expected a row-major tensor with unit inner stride")
    t2 = t.reshape(-1, t.shape[-1]) if t.is_contiguous() else t
    if t2.dim() != 2:
        raise _lib.BtError("non-contiguous input must be 2-D")
    return t2, t2.stride(0)


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
           residual: Optional[torch.Tensor] = None, act: str = "none", out: Optional[torch.Tensor] = None,
           scratch: Optional[torch.Tensor] = None):
    """out = act(x @ weight.T + bias) + residual   (weight in nn.Linear layout [out_features, in_features]).
    ``scratch`` (any dtype, >= 2*M*N*4 bytes) lets skinny problems use the split-K weight-streaming path."""
    x2, lda = _rows(x)
    M, K = x2.shape
    if weight.dim() == 3:  # Conv1d(kernel_size=1) weight [out, in, 1]
        weight = weight[:, :, 0]
    N = weight.shape[0]
    if weight.shape[1] != K or not weight.is_contiguous():
        raise _lib.BtError(f"weight must be contiguous [N, K={K}], got {tuple(weight.shape)}")
    if out is None:
        out = torch.empty(x.shape[:-1] + (N,), dtype=x.dtype, device=x.device)
    o2, ldo = _rows(out)
    ldr = 0
    if residual is not None:
        r2, ldr = _rows(residual)
    a = _lib.BT_ACT_GELU if act == "gelu" else _lib.BT_ACT_NONE
    if scratch is None:
        check(lib().bt_linear(dtype_code(x.dtype), M, N, K, ptr(x2), lda, ptr(weight), ptr(bias), ptr(residual), ldr,
                              ptr(o2), ldo, a, current_stream()), "bt_linear")
    else:
        check(lib().bt_linear_ws(dtype_code(x.dtype), M, N, K, ptr(x2), lda, ptr(weight), ptr(bias), ptr(residual), ldr,
                                 ptr(o2), ldo, a, ptr(scratch), scratch.numel() * scratch.element_size(),
                                 current_stream()), "bt_linear_ws")
    return out


def stream_k_scratch(device) -> torch.Tensor:
    """Scratch for the stream-K GEMM path of bt_linear_ws / bt_qkv_rope_append / bt_linear2_ws (hand-off slabs + flag words; the
    entry points clear the flag words themselves, the sticky time-out word behind them is zeroed here and read by
    ``stream_k_check``)."""
    scratch = torch.empty(int(lib().bt_stream_k_scratch_bytes()), dtype=torch.uint8, device=device)
    check(lib().bt_stream_k_reset(ptr(scratch), scratch.numel(), current_stream()), "bt_stream_k_reset")
    return scratch


def stream_k_check(scratch: torch.Tensor) -> None:
    """Raise if a stream-K launch on ``scratch`` flagged a hand-off time-out since the last reset (synchronises)."""
    err = torch.zeros(1, dtype=torch.int32, device=scratch.device)
    check(lib().bt_stream_k_status(ptr(scratch), scratch.numel() * scratch.element_size(), ptr(err), current_stream()),
          "bt_stream_k_status")
    if int(err.item()) & _lib.BT_DEV_ERR_HANDOFF_TIMEOUT:
        check(lib().bt_stream_k_reset(ptr(scratch), scratch.numel() * scratch.element_size(), current_stream()), "bt_stream_k_reset")
        raise _lib.BtError("stream-K GEMM hand-off timed out: the results of the launches on this scratch are incomplete")


def linear2(x1: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, x2: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor,
            residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, scratch: Optional[torch.Tensor] = None):
    """out = x1 @ w1.T + b1 + x2 @ w2.T + b2 + residual in one call (attention.dense + mlp.dense_4h_to_h of a parallel-residual
    layer); with ``scratch`` (stream_k_scratch) stacks above 256 rows run as ONE stream-K GEMM over K1 + K2."""
    a1, ld1 = _rows(x1)
    a2, ld2 = _rows(x2)
    M, K1 = a1.shape
    K2 = a2.shape[1]
    N = w1.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=x1.dtype, device=x1.device)
    o2, ldo = _rows(out)
    ldr = 0
    if residual is not None:
        _, ldr = _rows(residual)
    if scratch is None:
        check(lib().bt_linear2(dtype_code(x1.dtype), M, N, K1, ptr(a1), ld1, ptr(w1), ptr(b1), K2, ptr(a2), ld2, ptr(w2), ptr(b2),
                               ptr(residual), ldr, ptr(o2), ldo, current_stream()), "bt_linear2")
    else:
        check(lib().bt_linear2_ws(dtype_code(x1.dtype), M, N, K1, ptr(a1), ld1, ptr(w1), ptr(b1), K2, ptr(a2), ld2, ptr(w2), ptr(b2),
                                  ptr(residual), ldr, ptr(o2), ldo, ptr(scratch), scratch.numel() * scratch.element_size(),
                                  current_stream()), "bt_linear2_ws")
    return out


def layernorm(x, w1, b1, eps: float, w2=None, b2=None):
    """LN(x)*w1+b1 (and LN(x)*w2+b2 when w2 is given) in one pass."""
    x2, ldx = _rows(x)
    M, H = x2.shape
    o1 = torch.empty_like(x2 if x2.is_contiguous() else x2.contiguous())
    o2 = torch.empty_like(o1) if w2 is not None else None
    check(lib().bt_layernorm(dtype_code(x.dtype), M, H, ptr(x2), ldx, ptr(w1), ptr(b1), ptr(o1), H, ptr(w2), ptr(b2),
                             ptr(o2), H, float(eps), current_stream()), "bt_layernorm")
    o1 = o1.view(x.shape) if x.is_contiguous() else o1
    if o2 is None:
        return o1
    return o1, (o2.view(x.shape) if x.is_contiguous() else o2)


def embed_rows(ids: torch.Tensor, table: torch.Tensor, out: Optional[torch.Tensor] = None,
               row_zero: Optional[torch.Tensor] = None, rows_per_seq: int = 1):
    """out[r] = table[ids[r]] for the flattened ids (int64; contiguous, or 1-D with any stride — no copy is made)."""
    if ids.dtype != torch.int64:
        raise _lib.BtError("ids must be int64")
    if ids.dim() == 1:
        ids_stride = ids.stride(0) if ids.numel() > 1 else 1
    elif ids.is_contiguous():
        ids_stride = 1
    else:
        raise _lib.BtError("ids must be contiguous or 1-D strided")
    rows = ids.numel()
    vocab, width = table.shape
    if out is None:
        out = torch.empty(ids.shape + (width,), dtype=table.dtype, device=table.device)
    check(lib().bt_embed_rows(dtype_code(table.dtype), rows, width, ptr(ids), ids_stride, vocab, ptr(table), ptr(out),
                              width, ptr(row_zero), rows_per_seq, current_stream()), "bt_embed_rows")
    return out


def argmax(logits: torch.Tensor, out: Optional[torch.Tensor] = None):
    l2, ld = _rows(logits)
    M, V = l2.shape
    if out is None:
        out = torch.empty(M, dtype=torch.int64, device=logits.device)
    check(lib().bt_argmax(dtype_code(logits.dtype), M, V, ptr(l2), ld, ptr(out), out.stride(0) if out.dim() else 1,
                          current_stream()), "bt_argmax")
    return out


def logits_argmax(hid: torch.Tensor, weight: torch.Tensor, logits: torch.Tensor, store_logits: bool = True,
                  out: Optional[torch.Tensor] = None, ws: Optional[torch.Tensor] = None):
    """ids = argmax(hid @ weight.T) (greedy token step): the logits GEMM with the argmax in its epilogue where the persistent MFMA
    kernel applies (``logits`` is then only written when ``store_logits``), bt_linear + bt_argmax through ``logits`` otherwise."""
    h2, ldh = _rows(hid)
    M, K = h2.shape
    V = weight.shape[0]
    l2, ldl = _rows(logits)
    if out is None:
        out = torch.empty(M, dtype=torch.int64, device=hid.device)
    if ws is None:
        ws = torch.empty(max(int(lib().bt_logits_argmax_workspace_bytes(M, V)), 16), dtype=torch.uint8, device=hid.device)
    check(lib().bt_logits_argmax(dtype_code(hid.dtype), M, V, K, ptr(h2), ldh, ptr(weight), ptr(l2), ldl, int(bool(store_logits)),
                                 ptr(out), out.stride(0) if out.dim() else 1, ptr(ws), ws.numel() * ws.element_size(),
                                 current_stream()), "bt_logits_argmax")
    return out


def topk_sample(logits: torch.Tensor, temperature: float = 1.0, top_k: int = 50, seed: int = 0,
                step: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, row0: int = 0):
    l2, ld = _rows(logits)
    M, V = l2.shape
    if out is None:
        out = torch.empty(M, dtype=torch.int64, device=logits.device)
    check(lib().bt_topk_sample(dtype_code(logits.dtype), M, V, ptr(l2), ld, float(temperature), int(top_k), int(seed),
                               ptr(step), int(row0), ptr(out), out.stride(0), current_stream()), "bt_topk_sample")
    return out


def rope_tables(max_pos: int, rot: int, base: float, device) -> torch.Tensor:
    """fp32 [2, max_pos, rot/2] cos/sin tables, built on the host once (angle = pos * base^(-2j/rot))."""
    inv_freq = 1.0 / (base ** (torch.arange(0, rot, 2, dtype=torch.float32) / rot))
    freqs = torch.outer(torch.arange(max(max_pos, 1), dtype=torch.float32), inv_freq)
    return torch.stack([freqs.cos(), freqs.sin()]).contiguous().to(device)
