#!/bin/bash
# first_multi_gpu.sh -- ONE command that turns the first lease of a box with several MI355X into the complete evidence
# set for SURVEY section 8 rows A8 / (e) / (f2): everything that has only ever run with ranks SHARING one GPU.
#
#   bash scripts/first_multi_gpu.sh [--gpus N] [--shared] [--quick]
#
#   --gpus N   ranks to go up to (default: the GPUs the kernel driver lists; 2, 4, 8 <= N are benchmarked)
#   --shared   dry run on a one-GPU box: the same steps with the ranks sharing the device (flow check, not a measurement;
#              the pytest step then only shows its skip reasons)
#   --quick    smaller problem (256^3) and fewer cycles in the bench steps (dry runs)
#
# Steps, in order; every step is bounded by its own timeout and the script goes on if one fails:
#   1. inventory: GPUs, topology, RCCL / HIP versions, the MPI installations in reach and whether one of them is ROCm-aware
#   2. link probe of every ordered GPU pair, copy engines and compute-unit stores   (scripts/probe/link_matrix.py)
#   3. the un-shimmed multi-GPU tests: real RCCL between ranks, cross-device IPC, remote stores, autotuner with every
#      backend, config 5 halos -- every cell                                       (tests/test_gpu_multi_device.py)
#   3b. the device-pointer MPI path (CUDECOMP_MPI_GPU_AWARE=1; reference comm_routines.h:325-413, 708-735) when step 1 found a
#      ROCm-aware MPI: CUDECOMP_TEST_MPI_IS_GPU_AWARE=1 pytest tests/test_gpu_mpi_flavour.py (never executed anywhere so far)
#   4. bench.py --gpus 2 / 4 / 8 as the driver launches it: every candidate under config.also_measured, the xgmi block
#      with the MEASURED link rate
#   5. flags in device memory vs the host-pinned board: tiny-transpose latency (16^3 fp64 on 2 / 4 / 8 ranks, ONE RANK PER DEVICE:
#      the fused small exchange of the NVSHMEM enum and NVSHMEM_SM -- is the 80 us at 8 ranks of the shared-GPU runs the shared
#      device or the flag fan-out?) and the 1024^3 cycle over NVSHMEM_PL
#   5b. the two-hop relay on the 2 x N/2 pencil grid (BASELINE config 3's grid at N = 8), on and off
#   6. rocprofv3 kernel + memory-copy timeline of one staged NVSHMEM_PL cycle and one config-5-style halo trio
#   7. summary -> gpurun_out/first_multi_gpu/summary.json next to the model table of DESIGN.md section 7
#      (copy it to profiles/rNN_scale_<N>gpus.json)
# Everything lands under gpurun_out/first_multi_gpu/.
cd "$(dirname "$0")/.."
OUT=gpurun_out/first_multi_gpu
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
NGPU=$(python - <<'PY'
import glob
n = 0
for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
    try:
        for line in open(f):
            if line.startswith("simd_count") and int(line.split()[1]) > 0:
                n += 1
    except (OSError, ValueError):
        pass
print(n)
PY
)
MAXR=$NGPU; SHARED=0; QUICK=0
while [ $# -gt 0 ]; do
  case "$1" in
    --gpus) MAXR=$2; shift 2;;
    --shared) SHARED=1; shift;;
    --quick) QUICK=1; shift;;
    *) echo "unknown option $1"; exit 2;;
  esac
done
if [ $SHARED -eq 0 ] && [ $MAXR -gt $NGPU ]; then MAXR=$NGPU; fi
SIZE=1024; STEPS=5; WARM=3
if [ $QUICK -eq 1 ]; then SIZE=256; STEPS=2; WARM=1; fi
echo "first_multi_gpu: $NGPU GPU(s) listed, ranks up to $MAXR, shared=$SHARED, size=$SIZE" | tee $OUT/00_plan.txt
step() { echo "== $(date +%H:%M:%S) step $1" | tee -a $OUT/00_plan.txt; }

step "1 inventory"
{ echo "gpus: $NGPU"; (rocm-smi --showtopo 2>&1 || true) | head -60; (rocm-smi --showproductname 2>&1 || true) | head -30;
  python -c "import torch; print('torch', torch.__version__, 'hip', torch.version.hip, 'devices', torch.cuda.device_count())" 2>&1; } > $OUT/01_inventory.txt

step "1b MPI inventory"
MPI_AWARE=0
{ for m in mpirun mpiexec ompi_info mpichversion; do echo "$m: $(command -v $m || echo none)"; done
  (ompi_info 2>/dev/null | grep -i -E "rocm|accelerator|Open MPI:" || true)
  (ompi_info --parsable --all 2>/dev/null | grep -i "mpi_built_with_rocm_support" || true)
  (mpichversion 2>/dev/null | head -12 || true)
  (/opt/conda/bin/mpichversion 2>/dev/null | head -4 || true)
  (ucx_info -v 2>/dev/null | head -3; ucx_info -d 2>/dev/null | grep -i -c rocm | sed 's/^/ucx rocm transports: /' || true)
} > $OUT/01b_mpi_inventory.txt 2>&1
if ompi_info --parsable --all 2>/dev/null | grep -q "mpi_built_with_rocm_support:value:true"; then MPI_AWARE=1; fi
if mpichversion 2>/dev/null | grep -qi -E "hip|rocm"; then MPI_AWARE=1; fi
echo "ROCm-aware MPI found: $MPI_AWARE" | tee -a $OUT/01b_mpi_inventory.txt $OUT/00_plan.txt

step "2 link matrix"
timeout 600 python scripts/probe/link_matrix.py 256 > $OUT/02_link_matrix.json 2> $OUT/02_link_matrix.err || echo "link matrix failed (rc $?)" >> $OUT/00_plan.txt

step "3 multi-device tests"
( time timeout 3000 python -m pytest tests/test_gpu_multi_device.py -q -m gpu -rs --durations=10 ) > $OUT/03_multi_device_tests.log 2>&1
tail -3 $OUT/03_multi_device_tests.log | tee -a $OUT/00_plan.txt

step "3b device-pointer MPI path"
if [ $MPI_AWARE -eq 1 ]; then
  ( time CUDECOMP_TEST_MPI_IS_GPU_AWARE=1 timeout 1800 python -m pytest tests/test_gpu_mpi_flavour.py -q -m gpu -rs ) > $OUT/03b_mpi_gpu_aware.log 2>&1
  tail -3 $OUT/03b_mpi_gpu_aware.log | tee -a $OUT/00_plan.txt
else
  echo "   skipped: no ROCm-aware MPI in reach (see 01b_mpi_inventory.txt); the host-staged MPI path runs in the -m gpu suite" | tee -a $OUT/00_plan.txt $OUT/03b_mpi_gpu_aware.log
fi

step "4 bench.py at 2 / 4 / 8 ranks"
for n in 2 4 8; do
  [ $n -le $MAXR ] || continue
  # (as a plain command: bench.py starts its own ranks under torch.distributed.run when no launcher is around it -- the
  # way the driver called `--gpus 1`; the driver's own multi-GPU line with torch.distributed.run in front works alike)
  ( time timeout 1500 python bench.py --gpus $n --steps $STEPS --warmup $WARM --size $SIZE ) > $OUT/04_bench_n$n.log 2>&1
  grep -E '^\{' $OUT/04_bench_n$n.log | tail -1 > $OUT/04_bench_n$n.json
  echo "   n=$n: $(python -c "import json,sys; r=json.load(open('$OUT/04_bench_n$n.json')); print(r['ms_per_step'], 'ms per cycle,', r['config']['transport'], r['config']['pdims'])" 2>&1 | tail -1)" | tee -a $OUT/00_plan.txt
done

step "5 flags in device memory vs host-pinned board"
timeout 900 python scripts/probe/flag_latency.py > $OUT/05_flag_latency.json 2> $OUT/05_flag_latency.err || echo "flag latency failed" >> $OUT/00_plan.txt
for flags in 0 1; do
  n=$MAXR; [ $n -gt 8 ] && n=8
  [ $n -ge 2 ] || continue
  ( CUDECOMP_FLAGS_IN_DEVICE_MEMORY=$flags timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 \
      --master-port 2954$flags bench.py --gpus $n --steps $STEPS --warmup $WARM --size $SIZE --backend peer_pl --pdims 1 $n ) > $OUT/05_bench_peer_pl_flags$flags.log 2>&1
  grep -E '^\{' $OUT/05_bench_peer_pl_flags$flags.log | tail -1 > $OUT/05_bench_peer_pl_flags$flags.json
done

step "5b two-hop relay on a pencil grid (2 x N/2): NVSHMEM enum with and without CUDECOMP_TWO_HOP_RELAY"
for relay in 0 1; do
  n=$MAXR; [ $n -gt 8 ] && n=8
  [ $n -ge 4 ] || continue
  ( CUDECOMP_TWO_HOP_RELAY=$relay timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 \
      --master-port 2955$relay bench.py --gpus $n --steps $STEPS --warmup $WARM --size $SIZE --backend peer --pdims 2 $((n / 2)) ) > $OUT/05b_bench_2xN_relay$relay.log 2>&1
  grep -E '^\{' $OUT/05b_bench_2xN_relay$relay.log | tail -1 > $OUT/05b_bench_2xN_relay$relay.json
done

step "6 timelines (rocprofv3 kernel + memory-copy trace, no counters)"
( REPO=$PWD; cd /tmp && export TMPDIR=/tmp
  for engine in sdma cu; do
    OVERLAP_ENGINE=$engine timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $REPO/$OUT/06_prof/$engine -- \
      python $REPO/scripts/probe/overlap_run.py > $REPO/$OUT/06_prof_$engine.log 2>&1
  done
  cd $REPO; python scripts/summarize_overlap.py $OUT/06_prof > $OUT/06_timeline.txt 2> $OUT/06_timeline.err; cp $OUT/06_prof/summary.json $OUT/06_timeline.json
  find $OUT/06_prof -name "*_trace.csv" -delete; find $OUT/06_prof -name "*agent_info.csv" -delete ) || echo "timeline step failed" >> $OUT/00_plan.txt

step "7 summary"
python scripts/summarize_first_multi_gpu.py $OUT > $OUT/summary.json 2> $OUT/07_summary.err
python -c "import json; d=json.load(open('$OUT/summary.json')); print(json.dumps({k: d[k] for k in ('gpus','shared','bench')}, indent=1)[:3000])" 2>&1 | tee -a $OUT/00_plan.txt
echo "== done; copy $OUT/summary.json to profiles/rNN_scale_${MAXR}gpus.json" | tee -a $OUT/00_plan.txt
