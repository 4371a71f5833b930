"""Builds mmt_amd/lib/libmmt_hip.so from mmt_amd/csrc/*.hip with hipcc for gfx950.

hipcc cross-compiles without a GPU.  Objects are rebuilt only when a source or header is newer; the
shared library stays in-tree so that it travels to the GPU box with the repo snapshot.
"""
import concurrent.futures
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, 'lib', 'obj')
LIB = os.path.join(HERE, 'lib', 'libmmt_hip.so')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffast-math', '-fno-finite-math-only',
         '-Wall', '-Wno-unused-function', '-I' + os.path.join(ROOT, 'include')]


def _hipcc():
  return shutil.which('hipcc') or '/opt/rocm/bin/hipcc'


def _newest(paths):
  return max(os.path.getmtime(p) for p in paths)


def build(force=False, verbose=False, instr=False, lab=False):
  """The product library holds the tiles the dispatcher selects.  lab=True builds libmmt_hip_lab.so with every tile that was
  measured and lost as well (-DMMT_LAB_TILES: tools/gemm_lab.py, the MMT_TILE_* switches, the lab-only parity cases);
  instr=True builds libmmt_hip_instr.so = the lab tiles + s_memtime cycle counters inside the GEMM loops
  (tools/gemm_instr.py, tools/gemm2_budget.py, tools/g5_budget.py).  Use either through MMT_HIP_LIB=<path>."""
  global OBJ, LIB, FLAGS
  if instr:
    OBJ, LIB = os.path.join(HERE, 'lib', 'obj_instr'), os.path.join(HERE, 'lib', 'libmmt_hip_instr.so')
    FLAGS = FLAGS + ['-DMMT_GEMM2_INSTR', '-DMMT_G5_INSTR', '-DMMT_LAB_TILES'] + ['-D' + d for d in os.environ.get('MMT_LAB_DEFINES', '').split() if d]
  elif lab:
    OBJ, LIB = os.path.join(HERE, 'lib', 'obj_lab'), os.path.join(HERE, 'lib', 'libmmt_hip_lab.so')
    FLAGS = FLAGS + ['-DMMT_LAB_TILES'] + ['-D' + d for d in os.environ.get('MMT_LAB_DEFINES', '').split() if d]
  os.makedirs(OBJ, exist_ok=True)
  sources = sorted(glob.glob(os.path.join(CSRC, '*.hip')))
  headers = glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(ROOT, 'include', '*.h'))
  hdr_time = _newest(headers) if headers else 0.0
  jobs = []
  objs = []
  for src in sources:
    obj = os.path.join(OBJ, os.path.basename(src)[:-4] + '.o')
    objs.append(obj)
    stale = force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time)
    if stale:
      jobs.append([_hipcc()] + FLAGS + ['-c', src, '-o', obj])

  def run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    return cmd, r.returncode, r.stdout

  with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
    for cmd, rc, out in ex.map(run, jobs):
      if verbose or rc:
        print(' '.join(cmd))
        print(out)
      if rc:
        raise RuntimeError('hipcc failed for %s' % cmd[-3])
  if jobs or not os.path.exists(LIB):
    cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode:
      print(r.stdout)
      raise RuntimeError('link failed')
  return LIB


if __name__ == '__main__':
  print(build(force='--force' in sys.argv, verbose=True, instr='--instr' in sys.argv, lab='--lab' in sys.argv))
