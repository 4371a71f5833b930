"""Flat parameter storage for the native engine.

The HIP engine wants (a) the three Q/K/V weight matrices of a layer contiguous so that one fused GEMM
serves them, (b) gradients written straight into one fp32 buffer (one fused optimizer launch, one
RCCL all-reduce bucket), (c) bf16 shadow copies of the GEMM weights (plus W^T copies for the
input-gradient GEMMs).  At the same time the reference's state_dict names/shapes must stay intact
so released checkpoints load (base/base_trainer.py:430-432).

`FlatParams` therefore keeps ordinary `nn.Parameter`s -- registered under the reference's names by the
owning modules -- but re-points their `.data` at views of one flat fp32 tensor.  `nn.Module.to()` /
`load_state_dict` replace or overwrite `.data`; `ensure()` detects both (pointer check / version
counters) and re-flattens or re-packs lazily.
"""
import ctypes

import torch

from . import _lib
from ._lib import MmtPackItem, check

ALIGN = 64  # elements (256 B)


def _round_up(x, m):
  return (x + m - 1) // m * m


class FlatParams:

  def __init__(self, named_params):
    """named_params: ordered list of (name, nn.Parameter); order defines the flat layout."""
    self.names = [n for n, _ in named_params]
    self.params = [p for _, p in named_params]
    self.offsets = {}
    off = 0
    for p in self.params:
      self.offsets[id(p)] = off
      off += _round_up(p.numel(), ALIGN)
    self.count = _round_up(off, ALIGN)
    self.master = None
    self.grads = [None, None]
    self._which = 0
    self.shadows = []      # (key, src_param_list, rows, cols, dst_ld, dst, dst_t)
    self._shadow_by_key = {}
    self._packed_versions = None
    self._dirty = True  # set by writers that bypass torch's version counters (FlatAdam's raw-pointer kernel)
    self.device = None

  # ---- layout ----------------------------------------------------------------------------------
  def offset(self, p):
    return self.offsets[id(p)]

  def is_flat(self):
    if self.master is None:
      return False
    base = self.master.data_ptr()
    for p in self.params:
      if p.data_ptr() != base + 4 * self.offsets[id(p)] or p.device != self.master.device:
        return False
    return True

  def ensure(self, device):
    """Make every parameter a view of the flat master on `device` (values are preserved)."""
    device = torch.device(device)
    if self.master is not None and self.master.device == device and self.is_flat():
      return False
    master = torch.zeros(self.count, device=device, dtype=torch.float32)
    with torch.no_grad():
      for p in self.params:
        o = self.offsets[id(p)]
        view = master[o:o + p.numel()].view(p.shape)
        view.copy_(p.data.to(device=device, dtype=torch.float32))
        p.data = view
        p.grad = None
    self.master = master
    self.grads = [torch.zeros_like(master), None]
    self.device = device
    for sh in self.shadows:
      sh['dst'] = None
    self._packed_versions = None
    self._dirty = True
    return True

  def view(self, p, buf=None):
    buf = self.master if buf is None else buf
    o = self.offsets[id(p)]
    return buf[o:o + p.numel()].view(p.shape)

  def ptr(self, p, buf=None):
    buf = self.master if buf is None else buf
    return buf.data_ptr() + 4 * self.offsets[id(p)]

  def span(self, params):
    """(offset, count) of the contiguous run of the flat layout that holds exactly `params`."""
    ids = {id(p) for p in params}
    lo = min(self.offsets[i] for i in ids)
    hi = max(self.offsets[id(p)] + _round_up(p.numel(), ALIGN) for p in params)
    inside = [p for p in self.params if lo <= self.offsets[id(p)] < hi]
    if {id(p) for p in inside} != ids:
      raise ValueError('parameters are not one contiguous run of the flat layout')
    return lo, hi - lo

  def current_grad(self):
    """The buffer every backward function of the current step writes (chosen by select_grad_buffer)."""
    return self.grads[self._which]

  def select_grad_buffer(self):
    """Called once per forward (grad mode): pick a flat gradient buffer that no live `.grad` aliases, so
    that when the caller accumulates gradients over several backward passes autograd's `+=` adds two
    different buffers.  All backward functions of the step then use current_grad()."""
    cur = self.grads[self._which]
    lo, hi = cur.data_ptr(), cur.data_ptr() + 4 * self.count
    aliased = any(p.grad is not None and lo <= p.grad.data_ptr() < hi for p in self.params[:4] + self.params[-4:])
    if aliased:
      self._which ^= 1
      if self.grads[self._which] is None:
        self.grads[self._which] = torch.zeros_like(self.master)
    return self.grads[self._which]

  # ---- bf16 shadows ------------------------------------------------------------------------------
  def add_shadow(self, key, params, rows, cols, k_pad=None, transpose=False):
    """bf16 copy of the [rows, cols] matrix formed by the contiguous `params` (e.g. q,k,v weights).
    k_pad: leading dimension of the copy (zero padded).  transpose: also keep [k_pad_rows, rows]."""
    sh = dict(key=key, params=list(params), rows=rows, cols=cols, dst_ld=k_pad or cols, transpose=transpose,
              dst=None, dst_t=None)
    self.shadows.append(sh)
    self._shadow_by_key[key] = sh

  def shadow(self, key):
    sh = self._shadow_by_key[key]
    return sh['dst'], sh['dst_t']

  def _versions(self):
    return tuple(p._version for sh in self.shadows for p in sh['params'])

  def pack(self, force=False):
    """(Re)generate the bf16 shadows if any source weight changed since the last pack."""
    ver = self._versions()
    if not force and not self._dirty and ver == self._packed_versions and all(sh['dst'] is not None for sh in self.shadows):
      return False
    items = (MmtPackItem * len(self.shadows))()
    for i, sh in enumerate(self.shadows):
      first = sh['params'][0]
      o = self.offsets[id(first)]
      run = o
      for q in sh['params']:  # contiguity of fused blocks (q, k, v weights)
        assert self.offsets[id(q)] == run, 'fused shadow sources must be contiguous in the flat layout'
        run += q.numel()
      if sh['dst'] is None:
        sh['dst'] = torch.zeros(sh['rows'], sh['dst_ld'], device=self.device, dtype=torch.bfloat16)
        if sh['transpose']:
          sh['dst_t'] = torch.zeros(sh['dst_ld'], sh['rows'], device=self.device, dtype=torch.bfloat16)
      it = items[i]
      it.src = self.master.data_ptr() + 4 * o
      it.dst = sh['dst'].data_ptr()
      it.dst_t = sh['dst_t'].data_ptr() if sh['transpose'] else None
      it.rows, it.cols, it.dst_ld = sh['rows'], sh['cols'], sh['dst_ld']
      it.dst_t_ld, it.dst_t_rows = sh['rows'], sh['dst_ld']
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    check(_lib.lib().mmt_pack_weights(items, len(self.shadows), stream), 'mmt_pack_weights')
    self._packed_versions = ver
    self._dirty = False
    return True

  def shadows_fresh(self):
    """Called by an optimizer that rewrote the bf16 shadows together with the weights (FlatAdam's fused step)."""
    self._packed_versions = self._versions()
    self._dirty = False

  def adam_segments(self):
    """The flat buffer as an ascending list of segments that tile [0, count): ('plain', offset, count) spans and
    ('matrix', offset, shadow) entries, one per bf16 shadow (mmt_adam_step_fused).  None if a shadow cannot be
    refreshed by the optimizer (odd column count, not allocated yet)."""
    mats = []
    for sh in self.shadows:
      if sh['dst'] is None or sh['cols'] % 4 or (sh['transpose'] and sh['dst_t'] is None):
        return None
      mats.append((self.offsets[id(sh['params'][0])], sh))
    mats.sort(key=lambda t: t[0])
    segs, pos = [], 0
    for off, sh in mats:
      if off < pos:
        return None  # overlapping shadows: not a partition
      if off > pos:
        segs.append(('plain', pos, off - pos))
      segs.append(('matrix', off, sh))
      pos = off + sh['rows'] * sh['cols']
    if pos % 4:
      return None
    if pos < self.count:
      segs.append(('plain', pos, self.count - pos))
    return segs
