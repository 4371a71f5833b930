"""bench.py -- MMT hot-path training step on MI355X (driver contract: see the task statement).

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): MSRVTT jsfusion shape, 7 experts x 30
tokens, d512, 4 layers, 4 heads, I=3072, batch 32 pairs PER GPU (weak scaling), dropout 0.1, train mode.
One step = zero_grad + CENet forward (video side native; text heads on the synthetic text-tower output)
+ global-batch similarity + MaxMarginRankingLoss + backward + Adam step, inputs resident in HBM.
The text tower (HF bert-base, third party, out of scope) is replaced by synthetic (B,768) vectors.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from mmt_amd import dist as mdist  # noqa: E402
from mmt_amd import ops, synthetic  # noqa: E402
from mmt_amd.loss import MaxMarginRankingLoss  # noqa: E402
from mmt_amd.model import CENet, cross_view_similarity  # noqa: E402
from mmt_amd.feature_store import RaggedFeatures  # noqa: E402
from mmt_amd.train_step import FlatMinibatch, GraphedTrainStep  # noqa: E402

BF16_DENSE_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
BATCH, TOKENS, HIDDEN, LAYERS, HEADS, INTER, MAX_POS = 32, 30, 512, 4, 4, 3072, 32
# --config N = BASELINE.json configs[N] (single-GPU shapes; the default, 1, is the one the metric is quoted on)
CONFIGS = {
    1: dict(batch=32, tokens=30, hidden=512, layers=4, heads=4, inter=3072, max_pos=32,
            name='configs[1]: MSRVTT jsfusion shape, 7 experts x 30 tokens, d512, L4, H4, I3072'),
    3: dict(batch=32, tokens=100, hidden=512, layers=4, heads=4, inter=3072, max_pos=102,
            name='configs[3]: ActivityNet-style long sequences, 7 experts x 100 tokens (S = 708), d512, L4, H4, I3072'),
    4: dict(batch=128, tokens=30, hidden=1024, layers=6, heads=8, inter=6144, max_pos=32,
            name='configs[4] (per-rank encoder step): HowTo100M-scale synthetic, 7 experts x 30 tokens, d1024, L6, H8, I6144'),
}
WORKLOAD = CONFIGS[1]['name']


def select_config(n, dense=False):
  global BATCH, TOKENS, HIDDEN, LAYERS, HEADS, INTER, MAX_POS, WORKLOAD, PMC_CSV, STATS_CSV
  c = CONFIGS[n]
  # (--dense of configs[1] has its own passes: the packed shape's traffic beside dense algorithmic bytes is not evidence)
  key = 'dense' if (dense and n == 1) else n
  PMC_CSV = os.path.join('profiles', PMC_CSVS[key])
  STATS_CSV = os.path.join('profiles', STATS_CSVS[key])
  BATCH, TOKENS, HIDDEN, LAYERS, HEADS, INTER, MAX_POS = (c[k] for k in ('batch', 'tokens', 'hidden', 'layers', 'heads',
                                                                         'inter', 'max_pos'))
  WORKLOAD = c['name']


class SyntheticTextTower(torch.nn.Module):
  """Stands in for the out-of-scope HF text tower: returns the precomputed (B*C, 768) text vectors."""

  def __init__(self):
    super().__init__()
    self.config = type('C', (), {'hidden_size': 768})()
    self.embeddings = torch.nn.Module()
    self.text = None
    self.ignores_token_inputs = True  # CENet then skips building ids / masks / position ids for it

  def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, position_ids=None, head_mask=None):
    return (self.text[:, None, :],)


def build_model(pack, dropout=0.1, text_tower='synthetic'):
  vb = synthetic.vid_bert_params(hidden=HIDDEN, layers=LAYERS, heads=HEADS, inter=INTER, max_pos=MAX_POS, dropout=dropout)
  return CENet(l2renorm=False, expert_dims=synthetic.compute_dims(synthetic.MSRVTT_MODALITIES), tokenizer=None,
               keep_missing_modalities=True, test_caption_mode='indep', txt_inp='bertftn', txt_agg='bertftn',
               txt_wgh='emb', vid_wgh='none', vid_cont='bert', vid_inp='both', pos_enc='tint', out_tok='mxp',
               vid_bert_params=vb, txt_pro='gbn', same_dim=HIDDEN,
               txt_bert_params={'hidden_dropout_prob': dropout, 'attention_probs_dropout_prob': dropout},
               txt_bert=SyntheticTextTower() if text_tower == 'synthetic' else 'native', pack_tokens=pack)


def encoder_flops_per_step(batch, seq):
  """fwd = B*S*L*(8d^2 + 4dI + 4Sd); fwd+bwd = 3x (dense token count by convention)."""
  d, i = HIDDEN, INTER
  return 3.0 * batch * seq * LAYERS * (8 * d * d + 4 * d * i + 4 * seq * d)


class KernelProbe:
  """HIP events around one launch per step of a kernel family, recorded on the launch stream by the engine
  (mmt_probe_arm_site): site 0 = FFN up-projection GEMM, 1 = FFN down-projection GEMM, 2 = grouped weight gradients,
  3 / 4 = attention forward / backward."""

  def __init__(self, n, site=0):
    import ctypes

    from mmt_amd import _lib
    self.n, self.site = n, site
    self.start = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    self.stop = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    for e in self.start + self.stop:
      e.record()  # materialise the hipEvent_t handles
    torch.cuda.synchronize()
    self._a = (ctypes.c_void_p * n)(*[e.cuda_event for e in self.start])
    self._b = (ctypes.c_void_p * n)(*[e.cuda_event for e in self.stop])
    self._lib = _lib.lib()
    self._lib.mmt_probe_arm_site(site, self._a, self._b, n)

  def finish(self, stride=1, offset=0):
    """stride/offset: with the native text tower every step runs TWO encoder forwards (text first, then video): the
    video tower's launches are the odd ones."""
    used = self._lib.mmt_probe_count_site(self.site)
    self._lib.mmt_probe_arm_site(self.site, None, None, 0)
    ms = [self.start[i].elapsed_time(self.stop[i]) for i in range(offset, used, stride)]
    return sum(ms) / max(1, len(ms)) * 1e-3, len(ms)


# rocprofv3 evidence the JSON line quotes (tools/final_profiles.sh writes them): one set of PMC passes per shape -- the
# unpacked ('dense') timing of configs[1] has its own --, and the by-grid kernel-trace summary of the replayed graph
PMC_CSVS = {1: 'r06_pmc_kernels.csv', 3: 'r06_pmc_config3.csv', 4: 'r06_pmc_config4.csv', 'dense': 'r06_pmc_dense.csv'}
STATS_CSVS = {1: 'r06_kernel_stats_packed_by_grid.csv', 3: 'r06_kernel_stats_config3_by_grid.csv',
              4: 'r06_kernel_stats_config4_by_grid.csv', 'dense': 'r06_kernel_stats_dense_by_grid.csv'}
PMC_CSV = os.path.join('profiles', PMC_CSVS[1])
STATS_CSV = os.path.join('profiles', STATS_CSVS[1])


def pmc_traffic(kernel_subs, grid_sub=None):
  """HBM traffic per launch of a kernel from the committed rocprofv3 PMC passes (separate --pmc runs of this same command,
  tools/final_profiles.sh): (2 * FETCH_SIZE + WRITE_SIZE) KiB -- on gfx950 FETCH_SIZE reports half of a wide coalesced
  read (MI355X_MICROARCH.md, HBM section).  Of the grids a kernel was launched with, the one that holds most of its time
  (the full layers' launches, not the tail's).  The N = hidden epilogue kernel serves two GEMMs per layer (attention output,
  K = hidden, and FFN down-projection, K = intermediate) with one grid: its figure is the mean over both.  None if the
  profile is not there."""
  import csv
  path = os.path.join(ROOT, PMC_CSV)
  if not os.path.exists(path):
    return None
  best, best_time = None, -1.0
  with open(path) as f:
    for row in csv.DictReader(f):
      if any(k in row['kernel'] for k in kernel_subs) and (grid_sub is None or grid_sub in row['kernel']):
        try:
          t = float(row['dispatches']) * float(row['avg_ns'])
          if t > best_time:
            best, best_time = (2.0 * float(row['FETCH_SIZE']) + float(row['WRITE_SIZE'])) * 1024.0, t
        except (KeyError, ValueError):
          pass
  return best


def _blob(rel):
  import hashlib
  path = os.path.join(ROOT, rel)
  if not os.path.exists(path):
    return None
  data = open(path, 'rb').read()
  return hashlib.sha1(b'blob %d\0' % len(data) + data).hexdigest()


def pmc_blob():
  """git blob id of the PMC CSV the `traffic` figures come from (so a stale profile is visible in the JSON line)."""
  return _blob(PMC_CSV)


def graph_launch_us(kernel_subs, grid_sub=None):
  """Average duration of a kernel INSIDE the replayed step graph, from the committed rocprofv3 kernel-trace summary of this
  same command (rocpd_stats.py --by-grid --csv): the HIP-event probes of `avg_launch_us` time eager launches between two
  graphs, which run ~10 % longer than the same kernel as a node of the graph.  Of the grids a kernel ran with, the one that
  holds most of its time.  None if the profile is not there."""
  import csv
  path = os.path.join(ROOT, STATS_CSV)
  if not os.path.exists(path):
    return None
  best, best_time = None, -1.0
  with open(path) as f:
    for row in csv.DictReader(f):
      # (the PMC summary writes a launch geometry as '[grid 256 x 512]', the kernel-trace summary as '[256 x 512]')
      if any(k.replace('[grid ', '[') in row['Name'] for k in kernel_subs) and (grid_sub is None or grid_sub in row['Name']):
        try:
          t = float(row['TotalDurationNs'])
          if t > best_time:
            best, best_time = float(row['AverageNs']) * 1e-3, t
        except (KeyError, ValueError):
          pass
  return best


def site_roofline(site, rows, sec, used, sq_sum=0.0):
  """Roofline entry of a probed kernel family: algorithmic FLOPs of ONE launch at `rows` live token rows / its average
  HIP-event duration inside real training steps, against the dense bf16 MFMA peak (the more demanding roof: the
  algorithmic bytes of these GEMMs at ~5 TB/s would take 30-45 % of the measured time).  sq_sum: sum over the samples of
  (live sequence length)^2 (attention)."""
  d, i = HIDDEN, INTER
  if site in (3, 4):
    # QK^T and PV forward (4 s^2 d per sample); dP, dV, dS -> dQ, dK plus the recomputed QK^T backward (10 s^2 d).  Bytes:
    # fwd reads QKV, writes O + lse; bwd reads QKV, O-side quantities (dO, lse, delta), writes dQKV.
    fwd = site == 3
    name = ('self-attention forward (softmax(QK^T/sqrt(dh) + mask) -> dropout -> PV; attn_fwd_kernel, %d heads)' % HEADS if fwd else
            'self-attention backward (dQ, dK, dV with probabilities and dropout mask recomputed; attn_bwd_kernel, %d heads)' % HEADS)
    flops = (4.0 if fwd else 10.0) * sq_sum * d
    nbytes = rows * d * 2 * (4 if fwd else 7) + rows * HEADS * 4 * (1 if fwd else 2)
    subs, grid = (['attn_fwd_kernel'] if fwd else ['attn_bwd_kernel']), None
  elif site == 0:
    name, flops = 'FFN up-projection GEMM + bias + erf-GELU (N=%d, K=%d; gemm2_kernel, EPI BIAS_GELU)' % (i, d), 2.0 * rows * i * d
    nbytes = rows * d * 2 + i * d * 2 + 2 * rows * i * 2
    subs, grid = ['gemm2_kernel<256, 192, 4, 2, 2, 2', 'gemm2_kernel<128, 128, 2, 4, 2, 2', 'gemm3_kernel<2>', 'gemm5_kernel<2>', 'gemm5_kernel<2, 128>'], None
  elif site == 1:
    name, flops = 'FFN down-projection GEMM + bias + dropout + residual (N=%d, K=%d; gemm2_kernel<128,64> phased or gemm5_kernel, EPI BIAS_DROP_RES)' % (d, i), 2.0 * rows * d * i
    nbytes = rows * i * 2 + d * i * 2 + 2 * rows * d * 4
    subs, grid = ['gemm2_kernel<128, 64, 2, 2, 4, 3', 'gemm2_kernel<128, 64, 4, 2, 3, 3', 'gemm2_kernel<128, 128, 2, 4, 2, 3', 'gemm3_kernel<3>', 'gemm5_kernel<3, 128>', 'gemm5_kernel<3, 64>'], None
  else:
    name = 'grouped weight gradients of one encoder layer (dW1, dW2, dWqkv, dWo + bias gradients; wgrad_phased_kernel)'
    flops = 2.0 * rows * (2 * i * d + 4 * d * d)
    nbytes = rows * (2 * i + 6 * d) * 2 + (2 * i * d + 4 * d * d) * 4
    subs, grid = ['wgrad3_kernel', 'wgrad_phased_kernel [grid 256 ', 'wgrad_phased_kernel [grid 1024 ', 'wgrad_grouped_kernel'], None
  tf = flops / sec / 1e12
  g_us = graph_launch_us(subs, grid)
  graph = dict(avg_launch_us_graph=g_us, frac_graph=None, graph_source=STATS_CSV, graph_source_git_blob=_blob(STATS_CSV))
  if g_us:
    graph['frac_graph'] = ((nbytes / (g_us * 1e-6) / 8e12) if site in (3, 4) else (flops / (g_us * 1e-6) / 1e12 / BF16_DENSE_PEAK_TFLOPS))
  if site in (3, 4):
    # at ~110 live tokens per sample neither roof is near (a few % of the MFMA peak, ~15 % of HBM): the launch is bound by
    # its per-tile instruction stream and latency; priced against HBM, the nearer of the two roofs
    gbs = nbytes / sec / 1e9
    return dict(kernel=name, bound='hbm', achieved=gbs, peak=8000.0, unit='GB/s', frac=gbs / 8000.0, flops_per_launch=flops,
                mfma_frac=tf / BF16_DENSE_PEAK_TFLOPS, algorithmic_bytes_per_launch=nbytes, avg_launch_us=sec * 1e6,
                launches_timed=used, traffic=pmc_traffic(subs, grid),
                traffic_unit='bytes/launch (PMC: 2*FETCH_SIZE + WRITE_SIZE, %s)' % PMC_CSV, traffic_source_git_blob=pmc_blob(), **graph)
  return dict(kernel=name, bound='mfma', achieved=tf, peak=BF16_DENSE_PEAK_TFLOPS, unit='TFLOP/s',
              frac=tf / BF16_DENSE_PEAK_TFLOPS, flops_per_launch=flops, algorithmic_bytes_per_launch=nbytes,
              hbm_frac_of_8TBps=nbytes / sec / 8e12, avg_launch_us=sec * 1e6, launches_timed=used,
              traffic=pmc_traffic(subs, grid), traffic_unit='bytes/launch (PMC: 2*FETCH_SIZE + WRITE_SIZE, %s)' % PMC_CSV,
              traffic_source_git_blob=pmc_blob(), **graph)


def executed_flops_per_step(batch, live_rows, seq_lens_sq_sum, m_experts):
  """Encoder FLOPs actually executed with token packing + last-layer row elimination (fwd; x3 for fwd+bwd):
  L-1 full layers on the live rows, the last layer's QKV projection on the live rows and the rest of it on the B*M
  read-out rows only; attention over the live keys of every sample."""
  d, i = HIDDEN, INTER
  full = live_rows * (8 * d * d + 4 * d * i) + 4 * seq_lens_sq_sum * d
  tail_rows = batch * m_experts
  last = live_rows * 6 * d * d + tail_rows * (2 * d * d + 4 * d * i) + 4 * tail_rows * (live_rows / batch) * d
  return 3.0 * ((LAYERS - 1) * full + last)


def row_block_timing(b=8192, n=65536, m=7, d=1024, iters=3):
  """configs[4], second half: one rank's row block of the 64k-pair similarity + max-margin loss (mmt_amd/large_sim.py:
  8192 texts x 65536 videos, M = 7, d = 1024), forward + backward on this GPU, cross-rank quantities stood in by the
  block's own (parity at this size: tests/test_large_sim_gpu.py)."""
  from mmt_amd.large_sim import RowBlock
  dev = torch.device('cuda', torch.cuda.current_device())
  g = torch.Generator(device=dev).manual_seed(0)
  nrm = lambda x: torch.nn.functional.normalize(x, dim=-1)
  vid = nrm(torch.randn(n, m, d, device=dev, generator=g))
  txt = nrm(torch.randn(b, m, d, device=dev, generator=g) + 0.3 * vid[:b])
  tw = torch.softmax(torch.randn(b, m, device=dev, generator=g), -1)
  vw = torch.full((n, m), 1.0 / m, device=dev)
  best = None
  for _ in range(iters):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    blk = RowBlock(txt, tw, vid, vw, 0, 0.05)
    diag = torch.zeros(n, device=dev)
    diag[:b] = blk.phase_similarity()
    colcnt, loss = blk.phase_counts(diag)
    dtxt, dtw, q = blk.phase_backward(colcnt)
    blk.phase_video_grad(q[:b], vid[:b], vw[:b])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
  flops = 3 * 2.0 * b * n * m * d
  return dict(rows=b, cols=n, experts=m, dim=d, ms_fwd_bwd=best * 1e3, gemm_tflops=flops / best / 1e12,
              mfma_frac=flops / best / 1e12 / BF16_DENSE_PEAK_TFLOPS, loss=float(loss.item()))


def cpu_baseline(steps=6):
  """The CPU oracle ('port' of the reference path, pinned to it by tests/golden) on this box's host cores: one training
  step of config B as the reference runs it -- dense tokens, fp32, train mode with the dropout masks drawn inside the
  timed region (bernoulli, as ATen's dropout does: ~20 % of the reference's forward), backward,
  torch.optim.Adam step.  The reference itself (Python/torch) cannot travel to the GPU box; BASELINE.md section 4 holds
  its timing from the build container next to this port's timing on the same cores."""
  import copy

  from oracle import mmt_oracle as O
  torch.set_num_threads(min(32, os.cpu_count()))  # more threads thrash on these small ops (256-thread run: 0.28 pairs/s)
  p_drop = 0.1
  model = build_model(False, dropout=0.0)
  sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
  mods = list(synthetic.compute_dims(synthetic.MSRVTT_MODALITIES))
  cfg = dict(modalities=mods, expert_dims=synthetic.compute_dims(synthetic.MSRVTT_MODALITIES),
             vid_bert_params=synthetic.vid_bert_params(hidden=HIDDEN, layers=LAYERS, heads=HEADS, inter=INTER, dropout=p_drop),
             same_dim=HIDDEN)
  mb, text = synthetic.make_batch(0, BATCH, synthetic.MSRVTT_MODALITIES, TOKENS)
  seq = 1 + len(mods) * (TOKENS + 1)
  shapes = {'emb': (BATCH, seq, HIDDEN)}
  for l in range(LAYERS):
    shapes['l%d.probs' % l] = (BATCH, HEADS, seq, seq)
    shapes['l%d.attn_out' % l] = shapes['l%d.ffn_out' % l] = (BATCH, seq, HIDDEN)
  P = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point else v) for k, v in sd.items()}
  opt = torch.optim.Adam([v for v in P.values() if torch.is_tensor(v) and v.requires_grad], lr=5e-5)
  times = []
  for it in range(steps + 1):
    t0 = time.time()
    masks = {k: torch.bernoulli(torch.full(shp, 1.0 - p_drop)) for k, shp in shapes.items()}
    opt.zero_grad()
    sims = O.cenet_forward(P, cfg, copy.deepcopy(mb), text, training=True, masks=masks)['cross_view_conf_matrix']
    O.max_margin_ranking_loss(sims, 0.05, True).backward()
    opt.step()
    times.append(time.time() - t0)
  sec = sum(times[1:]) / steps
  cpu = ''
  try:
    with open('/proc/cpuinfo') as f:
      cpu = next(l.split(':', 1)[1].strip() for l in f if l.startswith('model name'))
  except (OSError, StopIteration):
    pass
  return dict(value=BATCH / sec, unit='pairs/s', cores=torch.get_num_threads(), kind='port', cpu=cpu,
              sample='%d training steps of config B (batch 32, dense, fp32, dropout 0.1 masks drawn per step, backward, '
                     'Adam; torch CPU ops; 1 warm-up step)' % steps)


def spawn_ranks(n):
  """Launcher for `python bench.py --gpus N` without torch.distributed.run: N child processes of this same command line
  with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment.  -> worst exit code."""
  import socket
  import subprocess
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  procs = []
  for r in range(n):
    env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
               MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')  # dmabuf IPC (RCCL across processes), see the task environment
    procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
  rc = 0
  try:
    while any(p.poll() is None for p in procs):
      time.sleep(0.2)
      if any(p.poll() not in (None, 0) for p in procs):
        break  # a rank that died takes the others down instead of leaving them in a collective forever
    rc = max(abs(p.poll() or 0) for p in procs)
  finally:
    for p in procs:
      if p.poll() is None:
        p.terminate()
        rc = rc or 1
  return rc


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--config', type=int, choices=sorted(CONFIGS), default=1,
                  help='BASELINE.json configs[N]: 1 = the headline MSRVTT shape (default), 3 = long sequences (S = 708), '
                       '4 = the d1024 / L6 encoder at batch 128 per rank (+ one rank\'s 8192 x 65536 similarity / loss row block)')
  ap.add_argument('--steps', type=int, default=50)
  ap.add_argument('--warmup', type=int, default=10)
  ap.add_argument('--dense', action='store_true', help='keep padded tokens (no variable-length packing)')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--grad-sync', choices=['auto', 'staged', 'single'], default='auto',
                  help='auto: staged backward with per-stage all-reduce when N > 1; staged/single force either')
  ap.add_argument('--text-tower', choices=['synthetic', 'native'], default='synthetic',
                  help='synthetic: (B,768) text vectors stand in for the text tower (the headline workload); native: '
                       'random-init bert-base-cased on the engine, token ids in, fine-tuned with the rest (SURVEY 8f.2)')
  ap.add_argument('--host-inputs', action='store_true',
                  help='minibatches live in pinned host memory: every step uploads one over PCIe (the PCIe-inclusive rate; '
                       'the headline value keeps the inputs resident in HBM)')
  ap.add_argument('--input-slots', type=int, default=0,
                  help='sets of static input buffers the step is captured for (GraphedTrainStep(input_slots=)); 0 = one per '
                       'resident minibatch (3 with --host-inputs); 1 = one set, every minibatch copied into it inside the step')
  ap.add_argument('--in-graph-feed', action='store_true',
                  help='with --host-inputs: the upload of minibatch i + 1 as a copy node of step i\'s graph (GraphedTrainStep(host_feed=)); '
                       'measured SLOWER than the copy-stream upload on this runtime (the node does not overlap the kernels)')
  ap.add_argument('--stream-wait-uploads', action='store_true',
                  help='with --host-inputs: order upload and step by a stream-side event wait (r03) instead of a host-side one')
  ap.add_argument('--ragged-inputs', action='store_true',
                  help='video features in the ragged bf16 wire format (mmt_amd.feature_store.RaggedFeatures: live rows only, '
                       'no cast kernel) instead of the reference\'s dict of dense fp32 tensors')
  ap.add_argument('--force-collectives', action='store_true',
                  help='N=1 only: run the all-gather / all-reduce plumbing on a 1-rank RCCL group (measures its overhead)')
  ap.add_argument('--capture-collectives', action='store_true',
                  help='EXPERIMENTAL: capture the RCCL collectives into the step graph (one graph launch per multi-rank step)')
  ap.add_argument('--eager', action='store_true', help='no HIP-graph capture (host-bound; for debugging)')
  ap.add_argument('--no-dense', action='store_true', help='skip the second (unpacked) timing of the same step')
  ap.add_argument('--grad-dtype', choices=['fp32', 'bf16'], default='fp32',
                  help='wire format of the gradient all-reduces at N > 1 (bf16: half the bytes over xGMI)')
  ap.add_argument('--grad-algo', choices=['allreduce', 'rs_ag'], default='allreduce',
                  help='N > 1: every gradient span as one all-reduce (RCCL picks ring / tree) or as reduce-scatter + '
                       'all-gather (the decomposition a direct full-mesh exchange over all xGMI links maps to)')
  ap.add_argument('--shard-optimizer', action='store_true',
                  help='N > 1 with --grad-algo rs_ag: Adam on the 1/N shard the reduce-scatter leaves on each rank, all-gather '
                       'of the updated weights instead of the reduced gradients (GraphedTrainStep(shard_optimizer=True))')
  ap.add_argument('--fill', type=float, default=None,
                  help='mean fraction of valid feature tokens in the synthetic minibatches (default: SURVEY 8d, U{0..30} '
                       'valid tokens per expert = 0.5); the headline number is quoted at the default')
  ap.add_argument('--fork', type=int, default=None,
                  help='bit mask of the work that leaves the main stream for a parallel branch of the step graph '
                       '(mmt_amd.train_step.FORK_*: 1 weight gradients, 2 ... in two early launches, 4 LN/table reductions, '
                       '16 per-region Adam, 32 text heads, 64 ReduceDim weight gradients); default 0 = one serial chain (forked graphs measured slower, DESIGN section 7)')
  ap.add_argument('--adam-riders', action='store_true',
                  help='one rank: the optimizer\'s units ride in the backward\'s GEMM launches (GraphedTrainStep(adam_riders=True); '
                       'measured slower than the one optimizer launch after the backward, DESIGN section 7)')
  ap.add_argument('--no-adam-riders', action='store_true', help='(the default; kept for the r06 A/B scripts)')
  ap.add_argument('--tower-base-ms', type=float, default=0.0,
                  help='with --text-tower native: ms/step of the SAME box\'s step with the synthetic tower, for the tower-only roofline')
  ap.add_argument('--comm-log', action='store_true',
                  help='N > 1: NCCL_DEBUG=INFO (RCCL prints the rings/trees and the algorithm + protocol of every collective)')
  args = ap.parse_args()
  select_config(args.config, args.dense)
  if args.config != 1:
    args.no_cpu_baseline = True  # the CPU port is timed on the headline shape only (a bounded sample of THAT workload)
  if args.ragged_inputs:
    if args.dense:
      ap.error('--ragged-inputs carries live rows only: it cannot feed the dense (unpacked) step')
    args.no_dense = True

  if args.capture_collectives and args.gpus > 1 and not os.environ.get('MMT_ALLOW_CAPTURED_COLLECTIVES'):
    # captured collectives have only ever run on a 1-rank communicator (which short-circuits RCCL's kernels): a multi-GPU
    # run must not pick an unvalidated path by accident
    raise SystemExit('--capture-collectives is validated on a 1-rank RCCL group only; refusing --gpus %d (set '
                     'MMT_ALLOW_CAPTURED_COLLECTIVES=1 to try it on a multi-GPU box)' % args.gpus)
  if args.shard_optimizer and args.grad_algo != 'rs_ag':
    ap.error('--shard-optimizer needs --grad-algo rs_ag')
  if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
    # bare `python bench.py --gpus N`: this process becomes the launcher -- N ranks of this same command, one per GPU,
    # rendezvous on 127.0.0.1 (what `python -m torch.distributed.run --nproc-per-node N` would set up); rank 0 prints
    # the ONE JSON line, the launcher only waits and forwards the worst exit code
    raise SystemExit(spawn_ranks(args.gpus))
  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if world != args.gpus:
    raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
  # Test hook (1-GPU boxes): MMT_BENCH_BACKEND=gloo runs every rank on cuda:0 with gloo moving the tensors, to exercise
  # the N > 1 control flow end to end where no second GPU exists.  The default is one GPU per rank over RCCL.
  backend = os.environ.get('MMT_BENCH_BACKEND', 'nccl')
  if backend != 'nccl':
    local_rank = 0
  torch.cuda.set_device(local_rank)
  dev = torch.device('cuda', local_rank)
  if world == 1 and args.force_collectives:
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
  if world > 1:
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if args.comm_log:
      os.environ['NCCL_DEBUG'] = 'INFO'
      os.environ.setdefault('NCCL_DEBUG_SUBSYS', 'INIT,COLL,TUNING')
    if backend == 'nccl':
      dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    else:
      dist.init_process_group(backend, rank=rank, world_size=world)

  loss_fn = MaxMarginRankingLoss(margin=0.05, fix_norm=True)
  seq = 1 + len(synthetic.MSRVTT_MODALITIES) * (TOKENS + 1)
  grad_dtype = torch.bfloat16 if args.grad_dtype == 'bf16' else None

  # NBATCH different synthetic minibatches resident in HBM (or in pinned host memory); the captured step exists once
  # per input slot and consumes a minibatch where it lies (--input-slots 1: each step copies one, device to device, into
  # the single set of static input buffers).
  NBATCH = 8 if args.config == 1 else 4
  batches, input_bytes = [], []
  for i in range(NBATCH):
    mb, text = synthetic.make_batch(1000 + 17 * rank + i, BATCH, synthetic.MSRVTT_MODALITIES, TOKENS, max_pos=MAX_POS,
                                    fill=args.fill)
    mb['text'] = text.view(-1, 768)
    if args.ragged_inputs:
      rag = RaggedFeatures.from_dense(mb['features'], mb['features_t'], mb['features_ind'], mb['features_maxpool'],
                                      experts=synthetic.MSRVTT_MODALITIES, pin_memory=args.host_inputs)
      input_bytes.append(rag.live_bytes())
      mb = {k: v for k, v in mb.items() if not k.startswith('features')}
      mb['features'] = rag
    else:
      input_bytes.append(sum(v.numel() * 4 for k in ('features', 'features_t', 'features_ind', 'features_maxpool')
                             for v in mb[k].values()))
    # one contiguous buffer per minibatch (HBM, or pinned host memory with --host-inputs): load = ONE copy
    batches.append(FlatMinibatch(mb, 'cpu', pin_memory=True) if args.host_inputs else FlatMinibatch(mb, dev))

  # packed token rows of every minibatch as a loader counts them on the host before the upload (CENet.count_live_rows; the
  # ragged wire format carries the count itself): the GEMM dispatcher prices a packed launch at its live size
  live_hints = [None] * NBATCH
  if not args.ragged_inputs:
    live_hints = [CENet.count_live_rows(b['features_ind']) for b in batches]
  text_hint = CENet.count_live_tokens(batches[0]['token_ids']) if args.text_tower == 'native' else None
  slots_used = 1
  in_graph_feed_used = False

  def timed_run(pack, steps, warmup):
    """Builds the model + captured step for one token layout and times `steps` steps as the contract prescribes
    (barrier + synchronize on both sides, MAX over ranks).  -> dict(model, runner, elapsed, first_loss, final_loss)"""
    torch.manual_seed(0)
    model = build_model(pack=pack, text_tower=args.text_tower).to(dev).train()
    model.text_live_rows_hint = text_hint  # (caption tokens of a minibatch, as a loader counts them: tile choice of the tower's GEMMs)
    mdist.broadcast_parameters(model)
    static = FlatMinibatch(batches[0], dev)

    def bind(st):  # the synthetic "text tower" hands out the minibatch's caption vectors: it holds a pointer to them
      if args.text_tower == 'synthetic':
        model.txt_bert.text = st['text']
    bind(static)
    # Input slots: the captured step exists once per set of input buffers, so a resident minibatch (or one that a copy
    # stream uploads while the previous step runs) is consumed where it lies.  --input-slots 1 = the r01-r02 arrangement:
    # ONE set of static inputs, every minibatch copied into it (device to device) inside the timed step.
    slots = 1 if args.eager else (args.input_slots or (NBATCH if args.in_graph_feed or not args.host_inputs else 3))
    # --host-inputs: the pinned minibatches ARE the loader's per-slot buffers; the captured step of slot s carries the
    # upload of slot s + 1 (GraphedTrainStep(host_feed=)).  --ragged-inputs keeps the r03 event-ordered upload path (its
    # copies depend on the live row counts of the minibatch).
    in_graph_feed = args.in_graph_feed and args.host_inputs and not args.ragged_inputs and not args.eager and slots == NBATCH and slots > 1
    runner = GraphedTrainStep(model, loss_fn, static, lr=5e-5, use_graphs=not args.eager,
                              overlap_grad_sync={'auto': None, 'staged': True, 'single': False}[args.grad_sync],
                              force_collectives=args.force_collectives, grad_dtype=grad_dtype,
                              capture_collectives=args.capture_collectives, fork=args.fork, grad_algo=args.grad_algo,
                              input_slots=slots, bind_inputs=bind, shard_optimizer=args.shard_optimizer,
                              host_feed=batches if in_graph_feed else None,
                              adam_riders=True if args.adam_riders else (False if args.no_adam_riders else None),
                              live_rows=live_hints[0] if pack else None)
    runner.measure_exposed = world > 1 or args.force_collectives
    runner.host_sync_uploads = not args.stream_wait_uploads
    nonlocal slots_used, in_graph_feed_used
    slots_used = slots
    in_graph_feed_used = in_graph_feed
    it, first = 0, None
    slot_batch = {}  # input slot -> index of the minibatch it holds
    if in_graph_feed:
      runner.prime(0)
      slot_batch.update({i: i % NBATCH for i in range(slots)})  # (slots == NBATCH: slot s always receives batches[s])

      def feed():  # nothing to do per step: step(slot) uploads slot + 1 from its pinned buffer inside its own graph
        nonlocal it
        cur = it % slots
        it += 1
        return cur
    elif slots > 1 and args.host_inputs:
      # minibatch i+1 crosses PCIe on a copy stream, straight into the next slot, while step i computes
      def feed():
        nonlocal it
        cur = it % slots
        it += 1
        runner.upload(batches[it % NBATCH], it % slots)
        slot_batch[it % slots] = it % NBATCH
        return cur
      runner.upload(batches[0], 0)
      slot_batch[0] = 0
    elif slots > 1:
      for i in range(slots):  # the resident minibatches ARE the slots' inputs
        runner.load(batches[i % NBATCH], i)
        slot_batch[i] = i % NBATCH

      def feed():
        nonlocal it
        cur = it % slots
        it += 1
        return cur
    elif args.host_inputs:
      # double-buffered upload: minibatch i+1 crosses PCIe on a copy stream while step i computes
      def feed():
        nonlocal it
        runner.load_prefetched()
        slot_batch[0] = it % NBATCH
        it += 1
        runner.prefetch(batches[it % NBATCH])
        return 0
      runner.prefetch(batches[0])
    else:
      def feed():
        nonlocal it
        runner.load(batches[it % NBATCH])
        slot_batch[0] = it % NBATCH
        it += 1
        return 0
    # every step passes the count of the minibatch it consumes, as a loader would: minibatches whose counts select other GEMM
    # tiles replay another capture of the step (GraphedTrainStep keeps one set of captures per tile choice)
    def hint(slot):
      return live_hints[slot_batch.get(slot, 0)] if pack else None
    for _ in range(warmup):
      s_ = feed()
      l = runner.step(s_, live_rows=hint(s_))
      if first is None:
        first = float(l.item())  # loss of the FIRST optimisation step (the runner's warm-up does not train)
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
      s_ = feed()
      loss = runner.step(s_, live_rows=hint(s_))
    torch.cuda.synchronize()
    if world > 1:
      dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
      t = torch.tensor([elapsed], device=dev if backend == 'nccl' else 'cpu', dtype=torch.float64)
      dist.all_reduce(t, op=dist.ReduceOp.MAX)
      elapsed = t.item()
    exposed = runner.exposed_collective_ms()
    runner._exposed = []
    return dict(model=model, runner=runner, elapsed=elapsed, first_loss=first, final_loss=float(loss.item()), it=it,
                exposed_ms=exposed)

  main_run = timed_run(not args.dense, args.steps, args.warmup)
  model, runner, elapsed, it = main_run['model'], main_run['runner'], main_run['elapsed'], main_run['it']
  main_first, main_final, exposed_ms = main_run['first_loss'], main_run['final_loss'], main_run['exposed_ms']

  # Kernel-family durations: HIP events around one launch per step of the three kernel families that lead the rocprof
  # time table, recorded by the engine on the launch stream in eager steps of the same workload right after the timed
  # region (events cannot be read back from inside a graph replay).
  probe_steps = 8
  towers = 2 if args.text_tower == 'native' else 1
  sites = [0, 1, 2, 3, 4] if towers == 1 else [0]
  probes = {st: KernelProbe(probe_steps * towers, st) for st in sites} if rank == 0 else {}
  live_rows, sq_sums = [], []
  plan0 = model._plans[next(iter(model._plans))]
  for _ in range(probe_steps):
    runner.load(batches[it % NBATCH]); it += 1
    runner.eager_step()
    live_rows.append(int(plan0.n_rows.item()))
    cu = plan0.cu.to(torch.float64)
    sq_sums.append(float(((cu[1:] - cu[:-1]) ** 2).sum().item()) if not args.dense else float(BATCH * seq * seq))
  torch.cuda.synchronize()
  site_times = {st: pr.finish(stride=towers, offset=towers - 1) for st, pr in probes.items()}
  staged = runner.staged
  # (parameters, parameters with bf16 shadows) of every flat buffer: the HBM accounting of --text-tower native
  tower_flat_sizes = [(int(o.flat.count), int(sum(sh['rows'] * sh['cols'] for sh in o.flat.shadows))) for o in runner.opt_flats]
  # share of the optimizer's units of work that rode in the backward's GEMM launches (the rest ran in the optimizer launch)
  riders = None
  if runner._rider_on:
    riders = []
    for o in runner.opt_flats:
      n_units, taken, nsteps = o.queue_stats()
      riders.append(dict(units=n_units, ridden_per_step=taken / max(nsteps, 1), ridden_fraction=taken / max(nsteps, 1) / n_units))
  del runner, model, main_run

  # the fill-independent figure in the same invocation: the same step WITHOUT token packing (every padded token computed)
  dense_run = None
  if not args.dense and not args.no_dense and world == 1 and not args.eager:
    dense_run = timed_run(False, max(10, args.steps // 2), max(3, args.warmup // 2))
    dense_ms = dense_run['elapsed'] / max(10, args.steps // 2) * 1e3
    dense_run = dict(ms_per_step=dense_ms, pairs_per_s=BATCH / dense_ms * 1e3)

  # every rank empties its C stdio buffer (RCCL's banner) BEFORE rank 0 prints, so the JSON line ends the job's stdout
  try:
    import ctypes
    ctypes.CDLL(None).fflush(None)
  except OSError:
    pass
  sys.stdout.flush()
  if world > 1:
    dist.barrier()
  if rank == 0:
    live = int(round(sum(live_rows) / len(live_rows)))  # mean live token rows per launch over the probe steps
    dense_rows = BATCH * seq
    pairs_per_s = world * BATCH * args.steps / elapsed
    flops = encoder_flops_per_step(BATCH, seq)
    rows = live if not args.dense else dense_rows
    executed = executed_flops_per_step(BATCH, rows, sum(sq_sums) / len(sq_sums), len(synthetic.MSRVTT_MODALITIES))
    out = {
        'metric': 'video-text pairs/sec (fwd+bwd+Adam), MSRVTT 7-expert d%d L%d' % (HIDDEN, LAYERS), 'value': pairs_per_s,
        'unit': 'pairs/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
        'config': {'workload': WORKLOAD + ', batch %d/GPU, dropout 0.1, train mode, Adam; ' % BATCH +
                               ('text tower replaced by synthetic (B,768) vectors' if args.text_tower == 'synthetic' else
                                'text tower = random-init bert-base-cased on the native engine, fine-tuned (30 tokens)'),
                   'global_batch': world * BATCH, 'seq_len': seq,
                   'parallelism': 'dp%d' % world, 'token_packing': not args.dense, 'hip_graphs': not args.eager,
                   'inputs': ('pinned host, uploaded every step (the copy of minibatch i + 1 is a node of step i\'s graph)' if in_graph_feed_used else
                              'pinned host, uploaded every step (double-buffered on a copy stream)') if args.host_inputs else 'resident in HBM',
                   # sets of input buffers the step was captured for (1 = every minibatch is copied, device to device, into
                   # one static set inside the timed step; > 1 = a minibatch is consumed in the slot it was put into)
                   'input_slots': slots_used,
                   'input_format': 'ragged bf16 wire buffer (live rows)' if args.ragged_inputs else 'dense fp32 dict',
                   'video_input_bytes_per_step': int(sum(input_bytes) / len(input_bytes)),
                   'text_tower': args.text_tower, 'grad_sync': 'staged' if staged else 'single',
                   'grad_wire_dtype': args.grad_dtype, 'grad_algo': args.grad_algo,
                   'optimizer_sharded': bool(args.shard_optimizer and world > 1), 'fill_arg': args.fill,
                   # one rank: per flat buffer, how much of the Adam step ran as riders of the backward's GEMM launches
                   'adam_riders': riders,
                   # mean ms per step between the end of the last backward stage and the last gradient reduction having
                   # landed (HIP events on the compute stream): what the staged overlap did not hide; null at N = 1
                   'exposed_collective_ms_rank0': exposed_ms,
                   'comm_backend': (backend if world > 1 else None), 'comm_log': bool(args.comm_log),
                   'collectives_captured': bool(args.capture_collectives and (world > 1 or args.force_collectives)),
                   'live_rows_rank0': live, 'dense_rows': dense_rows},
        # fraction of the (B, S) token grid that holds a real token in the synthetic batches (valid length ~ U{0..30}
        # per expert); the packed step computes only those, the dense step all of them.  The fill of real
        # MSRVTT features is unknown here (no dataset): `dense` below is the fill-independent number.
        'fill_fraction': live / dense_rows,
        'dense': dense_run,
        'encoder_dense_tflops': pairs_per_s / BATCH * flops / 1e12 / world,
        'encoder_dense_mfma_frac': pairs_per_s / BATCH * flops / 1e12 / world / BF16_DENSE_PEAK_TFLOPS,
        'executed_tflops': pairs_per_s / BATCH * executed / 1e12 / world,
        'executed_mfma_frac': pairs_per_s / BATCH * executed / 1e12 / world / BF16_DENSE_PEAK_TFLOPS,
        'first_loss': main_first, 'final_loss': main_final,
    }
    tops = [site_roofline(st, rows, sec, used, sum(sq_sums) / len(sq_sums)) for st, (sec, used) in sorted(site_times.items()) if used]
    # `roofline`: the family with the largest share of the step (each of them runs once per full layer, i.e.
    # LAYERS - 1 times per step with the compact last layer); `roofline_top3`: the three GEMM families and the two
    # attention launches (r04: the judge asked for attention next to them), same accounting, largest share first
    for r in tops:
      r['launches_per_step'] = LAYERS - 1
      r['share_of_step'] = (LAYERS - 1) * r['avg_launch_us'] * 1e-3 / (elapsed / args.steps * 1e3)
    tops.sort(key=lambda r: -r['share_of_step'])
    if tops:
      out['roofline'] = tops[0]
      out['roofline_top3'] = tops
      att = [r for r in tops if 'self-attention' in r['kernel']]
      if att:  # both attention launches of the full layers together, as a share of the step
        out['attention_share_of_step'] = sum(r['share_of_step'] for r in att)
    if args.text_tower == 'native':
      # The text tower at ~560 live caption tokens is a WEIGHT-STREAMING problem (its FLOPs, ~0.4 TF per step, are noise): the
      # step is priced against the HBM roof.  Algorithmic bytes per step = what must cross HBM once: every bf16 GEMM weight read by
      # the forward and its transposed copy by the input-gradient GEMMs, the fp32 gradient of every parameter written once, and
      # the Adam pass (g, w, m, v read; w, m, v + both bf16 shadows written) -- for BOTH flat buffers (video side + tower);
      # activations are negligible beside them.  `tower_only`: the same for the tower's buffer alone against the time the
      # tower adds to the step (this invocation's step minus the synthetic-tower step of the same box, when --tower-base-ms
      # gives it).
      def flat_bytes(n_params, n_shadowed):
        return dict(weights_bf16_fwd=2 * n_shadowed, weights_bf16_dgrad=2 * n_shadowed, grad_write_f32=4 * n_params,
                    adam_read=16 * n_params, adam_write=12 * n_params + 4 * n_shadowed)
      parts = []
      for nf in tower_flat_sizes:
        parts.append(flat_bytes(*nf))
      total = sum(sum(p.values()) for p in parts)
      sec = elapsed / args.steps
      out['roofline_hbm_step'] = dict(bound='hbm', unit='GB/s', peak=8000.0, achieved=total / sec / 1e9, frac=total / sec / 8e12,
                                      algorithmic_bytes_per_step=total, per_flat_buffer=parts,
                                      note='whole step (video side + native text tower); the tower alone: see tower_only')
      if args.tower_base_ms:
        tsec = sec - args.tower_base_ms * 1e-3
        tb = sum(parts[-1].values())
        out['roofline_hbm_step']['tower_only'] = dict(added_ms=tsec * 1e3, algorithmic_bytes=tb, achieved=tb / tsec / 1e9,
                                                      frac=tb / tsec / 8e12, hbm_floor_ms=tb / 6.3e12 * 1e3)
    if world == 1 and not args.no_cpu_baseline:
      out['cpu_baseline'] = cpu_baseline()
    if args.config == 4:
      out['similarity_loss_row_block'] = row_block_timing()
    # RCCL prints its version banner through C stdio, which is block-buffered on a pipe and would otherwise come out at
    # exit, AFTER the result: flush it first so that the JSON line is the last line of stdout
    try:
      import ctypes
      ctypes.CDLL(None).fflush(None)
    except OSError:
      pass
    print(json.dumps(out), flush=True)
  if dist.is_initialized():
    # ranks leave together and with an idle device: no communicator is torn down under a peer's pending kernel
    dist.barrier()
    torch.cuda.synchronize()
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
