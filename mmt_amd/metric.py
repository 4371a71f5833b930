"""Drop-in for the reference's `model/metric.py` retrieval metrics, computed on the MI355X.

`t2v_metrics(sims, query_masks=None)` / `v2t_metrics(sims, query_masks=None)` keep the reference signatures and
result keys (model/metric.py:26-150, 153-243, 246-258) but rank on the device: `sims` may be a CUDA tensor (it then
never leaves HBM -- only the n rank values are copied to the host) or a numpy array (uploaded once).
`retrieval_metrics(...)` goes one step further and also builds the N_text x N_video similarity on the device from
the gathered embeddings (the reference does both on the CPU: trainer/trainer.py:396-447).  SURVEY.md section 8f.1.
"""
import numpy as np
import torch

from . import _lib, ops
from ._lib import check


def _as_cuda_f32(x):
  if not torch.is_tensor(x):
    x = torch.from_numpy(np.ascontiguousarray(x))
  if not torch.cuda.is_available():
    raise RuntimeError('mmt_amd.metric needs a GPU (no CPU fallback)')
  return x.to(device='cuda', dtype=torch.float32).contiguous()


def retrieval_ranks(sims, query_masks=None):
  """sims [NQ = NV*cpv, NV] (rows = text queries) -> (t2v 0-based ranks [NQ], v2t best rank per video [NV]),
  tie-averaged like the reference; both float32 CUDA tensors."""
  sims = _as_cuda_f32(sims)
  nq, nv = sims.shape
  qm = None
  if query_masks is not None:
    qm = torch.as_tensor(np.asarray(query_masks.cpu() if torch.is_tensor(query_masks) else query_masks)).reshape(-1)
    qm = (qm != 0).to(device=sims.device, dtype=torch.uint8).contiguous()
    assert qm.numel() == nq
  t2v = torch.empty(nq, device=sims.device, dtype=torch.float32)
  v2t = torch.empty(nv, device=sims.device, dtype=torch.float32)
  scratch = torch.empty(nq, device=sims.device, dtype=torch.float32)
  check(_lib.lib().mmt_retrieval_ranks(ops._p(sims), ops._p(qm), nq, nv, ops._p(t2v), ops._p(v2t), ops._p(scratch),
                                       ops._stream()), 'mmt_retrieval_ranks')
  return t2v, v2t, qm


def cols2metrics(cols, num_queries):
  """model/metric.py:246-258 (same keys; numpy on the O(n) rank vector)."""
  cols = np.asarray(cols, dtype=np.float64)
  metrics = {}
  metrics['R1'] = 100 * float(np.sum(cols == 0)) / num_queries
  metrics['R5'] = 100 * float(np.sum(cols < 5)) / num_queries
  metrics['R10'] = 100 * float(np.sum(cols < 10)) / num_queries
  metrics['R50'] = 100 * float(np.sum(cols < 50)) / num_queries
  metrics['MedR'] = float(np.median(cols) + 1)
  metrics['MeanR'] = float(np.mean(cols) + 1)
  stats = np.array([metrics[x] for x in ('R1', 'R5', 'R10')])
  metrics['geometric_mean_R1-R5-R10'] = float(np.exp(np.mean(np.log(stats)))) if (stats > 0).all() else 0.0
  return metrics


def t2v_metrics(sims, query_masks=None):
  """Text-to-video retrieval metrics; sims: (N_text, N_video), text rows grouped per video."""
  t2v, _, qm = retrieval_ranks(sims, query_masks)
  cols = t2v.cpu().numpy()
  if qm is not None:
    cols = cols[qm.cpu().numpy().astype(bool)]
  out = cols2metrics(cols, cols.size)
  out['cols'] = cols
  return out


def v2t_metrics(sims, query_masks=None):
  """Video-to-text retrieval metrics (best rank among a video's own captions)."""
  _, v2t, _ = retrieval_ranks(sims, query_masks)
  cols = v2t.cpu().numpy()
  out = cols2metrics(cols, cols.size)
  out['cols'] = cols
  return out


def eval_similarity(vid_embds, text_embds, vid_weights, text_weights):
  """(B,M,d), (B,M,C,d), (B,M), (B,C,M) -> sims (B*C, B) on the device ('indep' caption mode, model.py:826-836)."""
  vid = _as_cuda_f32(vid_embds)
  txt4 = _as_cuda_f32(text_embds)
  b, m, d = vid.shape
  c = txt4.shape[2]
  txt = txt4.permute(0, 2, 1, 3).reshape(b * c, m, d).contiguous()
  tw = _as_cuda_f32(text_weights).reshape(b * c, m).contiguous()
  vw = _as_cuda_f32(vid_weights).reshape(b, m).contiguous()
  L = _lib.lib()
  ws = torch.empty(L.mmt_sims_eval_workspace_floats(b * c, b, m, d), device=vid.device, dtype=torch.float32)
  sims = torch.empty(b * c, b, device=vid.device, dtype=torch.float32)
  check(L.mmt_sims_eval(ops._p(txt), ops._p(vid), ops._p(tw), ops._p(vw), b * c, b, m, d, ops._p(ws), ops._p(sims),
                        ops._stream()), 'mmt_sims_eval')
  return sims


def retrieval_metrics(vid_embds, text_embds, vid_weights, text_weights, query_masks=None):
  """Embeddings of the whole eval set -> {'t2v_metrics': {...}, 'v2t_metrics': {...}} without an n^2 host copy."""
  sims = eval_similarity(vid_embds, text_embds, vid_weights, text_weights)
  return {'t2v_metrics': t2v_metrics(sims, query_masks), 'v2t_metrics': v2t_metrics(sims, query_masks)}
