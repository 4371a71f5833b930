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
from mmt_amd.train_step import FlatMinibatch, GraphedTrainStep  # noqa: E402

BF16_DENSE_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
BATCH, TOKENS, HIDDEN, LAYERS, HEADS, INTER = 32, 30, 512, 4, 4, 3072


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
  vb = synthetic.vid_bert_params(hidden=HIDDEN, layers=LAYERS, heads=HEADS, inter=INTER, max_pos=32, dropout=dropout)
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
  """HIP events around the dominant kernel's launch inside the timed steps (engine hook mmt_probe_arm)."""

  def __init__(self, n):
    import ctypes

    from mmt_amd import _lib
    self.n = n
    self.start = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    self.stop = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    for e in self.start + self.stop:
      e.record()  # materialise the hipEvent_t handles
    torch.cuda.synchronize()
    self._a = (ctypes.c_void_p * n)(*[e.cuda_event for e in self.start])
    self._b = (ctypes.c_void_p * n)(*[e.cuda_event for e in self.stop])
    self._lib = _lib.lib()
    self._lib.mmt_probe_arm(self._a, self._b, n)

  def finish(self, stride=1, offset=0):
    """stride/offset: with the native text tower every step runs TWO encoder forwards (text first, then video): the
    video tower's launches are the odd ones."""
    used = self._lib.mmt_probe_count()
    self._lib.mmt_probe_arm(None, None, 0)
    ms = [self.start[i].elapsed_time(self.stop[i]) for i in range(offset, used, stride)]
    return sum(ms) / max(1, len(ms)) * 1e-3, len(ms)


def time_dominant_kernel(rows, iters=40):
  """Dominant kernel = the bf16 MFMA NT GEMM (FFN up-projection shape: rows x 3072 x 512, bias+GELU
  epilogue).  Timed alone with HIP events on the launch stream; algorithmic flops = 2*rows*I*d."""
  dev = torch.device('cuda')
  R = ops.pad_rows(rows)
  a = (torch.randn(R, HIDDEN, device=dev)).to(torch.bfloat16)
  w = (torch.randn(INTER, HIDDEN, device=dev) * 0.05).to(torch.bfloat16)
  bias = torch.randn(INTER, device=dev)
  out = torch.empty(R, INTER, device=dev, dtype=torch.bfloat16)
  out2 = torch.empty_like(out)
  for _ in range(5):
    ops.gemm_nt(a, w, out, 'BIAS_GELU', m=rows, bias=bias, out2=out2)
  torch.cuda.synchronize()
  s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  s.record()
  for _ in range(iters):
    ops.gemm_nt(a, w, out, 'BIAS_GELU', m=rows, bias=bias, out2=out2)
  e.record()
  torch.cuda.synchronize()
  sec = s.elapsed_time(e) * 1e-3 / iters
  flops = 2.0 * rows * INTER * HIDDEN
  return dict(bound='mfma', achieved=flops / sec / 1e12, peak=BF16_DENSE_PEAK_TFLOPS, unit='TFLOP/s',
              frac=flops / sec / 1e12 / BF16_DENSE_PEAK_TFLOPS, traffic=None,
              kernel='gemm2_kernel<128,128,2x4 waves,NS2,BIAS_GELU> rows=%d N=%d K=%d' % (rows, INTER, HIDDEN),
              avg_launch_us=sec * 1e6)


def pmc_traffic(kernel_sub='gemm2_kernel<128, 128, 2, 4, 2, 2', grid_sub='[grid 1320 '):
  """HBM traffic of the dominant kernel per launch from the committed rocprofv3 PMC passes (profiles/r01_pmc_kernels.csv,
  separate --pmc runs of this same command): (2 * FETCH_SIZE + WRITE_SIZE) KiB -- on gfx950 FETCH_SIZE reports half of a
  wide coalesced read (MI355X_MICROARCH.md, HBM section).  None if the profile is not there."""
  import csv
  path = os.path.join(ROOT, 'profiles', 'r01_pmc_kernels.csv')
  if not os.path.exists(path):
    return None
  with open(path) as f:
    for row in csv.DictReader(f):
      if kernel_sub in row['kernel'] and grid_sub in row['kernel']:
        try:
          return (2.0 * float(row['FETCH_SIZE']) + float(row['WRITE_SIZE'])) * 1024.0
        except (KeyError, ValueError):
          return None
  return None


def cpu_baseline(steps=6):
  """The CPU oracle ('port' of the reference path, pinned to it by tests/golden) on this box's host cores:
  fwd+bwd of config B, train mode semantics without dropout RNG (cheaper than the reference), fp32."""
  import copy

  from oracle import mmt_oracle as O
  torch.set_num_threads(min(32, os.cpu_count()))  # more threads thrash on these small ops (256-thread run: 0.28 pairs/s)
  model = build_model(False, dropout=0.0)
  sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
  mods = list(synthetic.compute_dims(synthetic.MSRVTT_MODALITIES))
  cfg = dict(modalities=mods, expert_dims=synthetic.compute_dims(synthetic.MSRVTT_MODALITIES),
             vid_bert_params=synthetic.vid_bert_params(hidden=HIDDEN, layers=LAYERS, heads=HEADS, inter=INTER, dropout=0.0),
             same_dim=HIDDEN)
  mb, text = synthetic.make_batch(0, BATCH, synthetic.MSRVTT_MODALITIES, TOKENS)
  times = []
  for it in range(steps + 1):
    P = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point else v) for k, v in sd.items()}
    t0 = time.time()
    sims = O.cenet_forward(P, cfg, copy.deepcopy(mb), text, training=True)['cross_view_conf_matrix']
    O.max_margin_ranking_loss(sims, 0.05, True).backward()
    times.append(time.time() - t0)
  sec = sum(times[1:]) / steps
  return dict(value=BATCH / sec, unit='pairs/s', cores=torch.get_num_threads(), kind='port',
              sample='%d fwd+bwd steps of config B (batch 32, dense, fp32, torch CPU ops, 1 warm-up)' % steps)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
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
  ap.add_argument('--force-collectives', action='store_true',
                  help='N=1 only: run the all-gather / all-reduce plumbing on a 1-rank RCCL group (measures its overhead)')
  ap.add_argument('--eager', action='store_true', help='no HIP-graph capture (host-bound; for debugging)')
  args = ap.parse_args()

  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if world != args.gpus:
    raise SystemExit('--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)' % (args.gpus, world))
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
    if backend == 'nccl':
      dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    else:
      dist.init_process_group(backend, rank=rank, world_size=world)

  torch.manual_seed(0)
  model = build_model(pack=not args.dense, text_tower=args.text_tower).to(dev).train()
  mdist.broadcast_parameters(model)
  loss_fn = MaxMarginRankingLoss(margin=0.05, fix_norm=True)

  # NBATCH different synthetic minibatches resident in HBM; each step copies one (device-to-device) into
  # the static input buffers of the captured graphs.
  NBATCH = 16
  batches = []
  for i in range(NBATCH):
    mb, text = synthetic.make_batch(1000 + 17 * rank + i, BATCH, synthetic.MSRVTT_MODALITIES, TOKENS)
    mb['text'] = text.view(-1, 768)
    # one contiguous buffer per minibatch (HBM, or pinned host memory with --host-inputs): load = ONE copy
    batches.append(FlatMinibatch(mb, 'cpu', pin_memory=True) if args.host_inputs else FlatMinibatch(mb, dev))
  static = FlatMinibatch(batches[0], dev)
  if args.text_tower == 'synthetic':
    model.txt_bert.text = static['text']
  seq = 1 + len(synthetic.MSRVTT_MODALITIES) * (TOKENS + 1)
  runner = GraphedTrainStep(model, loss_fn, static, lr=5e-5, use_graphs=not args.eager,
                            overlap_grad_sync={'auto': None, 'staged': True, 'single': False}[args.grad_sync],
                            force_collectives=args.force_collectives)
  it = 0
  first_loss = None
  for _ in range(args.warmup):
    runner.load(batches[it % NBATCH]); it += 1
    l = runner.step()
    if first_loss is None:
      first_loss = float(l.item())  # loss of the first replayed step (after the runner's own eager warm-up steps)
  if world > 1:
    dist.barrier()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    runner.load(batches[it % NBATCH]); it += 1
    loss = runner.step()
  torch.cuda.synchronize()
  if world > 1:
    dist.barrier()
  elapsed = time.perf_counter() - t0
  if world > 1:
    t = torch.tensor([elapsed], device=dev if backend == 'nccl' else 'cpu', dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = t.item()
  final_loss = float(loss.item())

  # dominant-kernel duration: HIP events around its launch (engine probe) in eager steps of the same
  # workload right after the timed region (events cannot be read back from inside a graph replay)
  probe_steps = 8
  towers = 2 if args.text_tower == 'native' else 1
  probe = KernelProbe(probe_steps * towers) if rank == 0 else None
  live_rows = []
  plan0 = model._plans[next(iter(model._plans))]
  for _ in range(probe_steps):
    runner.load(batches[it % NBATCH]); it += 1
    runner._eager_step()
    live_rows.append(int(plan0.n_rows.item()))
  torch.cuda.synchronize()

  if rank == 0:
    live = int(round(sum(live_rows) / len(live_rows)))  # mean live token rows per launch over the probe steps
    pairs_per_s = world * BATCH * args.steps / elapsed
    flops = encoder_flops_per_step(BATCH, seq)
    out = {
        'metric': 'video-text pairs/sec (fwd+bwd+Adam), MSRVTT 7-expert d512 L4', 'value': pairs_per_s,
        'unit': 'pairs/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
        'config': {'workload': 'configs[1]: MSRVTT jsfusion shape, 7 experts x 30 tokens, d512, L4, H4, I3072, '
                               'batch 32/GPU, dropout 0.1, train mode, Adam; ' +
                               ('text tower replaced by synthetic (B,768) vectors' if args.text_tower == 'synthetic' else
                                'text tower = random-init bert-base-cased on the native engine, fine-tuned (30 tokens)'),
                   'global_batch': world * BATCH, 'seq_len': seq,
                   'parallelism': 'dp%d' % world, 'token_packing': not args.dense, 'hip_graphs': not args.eager, 'inputs': 'pinned host, uploaded every step' if args.host_inputs else 'resident in HBM', 'text_tower': args.text_tower, 'grad_sync': 'staged' if runner.staged else 'single',
                   'live_rows_rank0': live, 'dense_rows': BATCH * seq},
        'encoder_dense_tflops': pairs_per_s / BATCH * flops / 1e12 / world,
        'encoder_dense_mfma_frac': pairs_per_s / BATCH * flops / 1e12 / world / BF16_DENSE_PEAK_TFLOPS,
        'first_loss': first_loss, 'final_loss': final_loss,
    }
    rows = live if not args.dense else BATCH * seq
    sec, used = probe.finish(stride=towers, offset=towers - 1)
    kflops = 2.0 * rows * INTER * HIDDEN
    alone = time_dominant_kernel(rows)
    out['roofline'] = dict(bound='mfma', achieved=kflops / sec / 1e12, peak=BF16_DENSE_PEAK_TFLOPS, unit='TFLOP/s',
                           frac=kflops / sec / 1e12 / BF16_DENSE_PEAK_TFLOPS,
                           traffic=pmc_traffic() if not args.dense else None,
                           traffic_unit='bytes/launch (PMC: 2*FETCH_SIZE + WRITE_SIZE, profiles/r01_pmc_kernels.csv)',
                           kernel=alone['kernel'], avg_launch_us=sec * 1e6, launches_timed=used,
                           standalone_us=alone['avg_launch_us'])
    if world == 1 and not args.no_cpu_baseline:
      out['cpu_baseline'] = cpu_baseline()
    print(json.dumps(out))
  if dist.is_initialized():
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
