#!/bin/bash
# code_size_bisect.sh build|run -- which property of the library's device code flips the "14 ms per synchronous null-stream
# operation" regime (profiles/r04_tuning.md, r05_code_size.md)?  Library variants, all with the DEFAULT kernels:
#   sep_few    + a separate translation unit (= a separate code object in .hip_fatbin) of 26 large never-launched kernels, +0.2 MB
#   sep_many   + a separate code object of 118 small never-launched kernels, +0.2 MB
#   same_tu    filler compiled INTO the code object of the row-copy kernels (ONE code object of 0.72 MB; in round 5's first run
#              of this script, before the kernels were split over several code objects, it was kernels.hip + 0.2 MB)
#   sep_small  + a separate code object, +0.07 MB (total 0.59 MB)
#   sep_mid    + a separate code object, +0.13 MB (total 0.65 MB)
# plus: the default library, the TUNING_VARIANTS library (0.72 MB of real kernels), that library with the start-up link probe
# skipped / restricted to one engine, and the default library under a test program that carries +0.2 MB of its own device code.
cd "$(dirname "$0")/../.."
ROCM=${ROCM:-/opt/rocm}
B=scripts/probe/bisect_libs
CXXFLAGS="-O3 -std=c++17 -fPIC -Iinclude -Icudecomp_amd/csrc -I$ROCM/include -D__HIP_PLATFORM_AMD__ --offload-arch=gfx950"
LDFLAGS="-shared -L$ROCM/lib -lrccl -lamdhip64 -lpthread -ldl -Wl,-rpath,$ROCM/lib -Wl,-soname,libcudecomp.so.0"
if [ "$1" = build ]; then
  make -s -C cudecomp_amd -j8 && make -s -C cudecomp_amd -j8 TUNING_VARIANTS=1 || exit 1
  OBJS=$(ls cudecomp_amd/build/*.o | grep -v kernels_rows.hip.o)
  variant() {  # name NK BODY same_tu
    mkdir -p $B/$1
    if [ "$4" = 1 ]; then
      $ROCM/bin/hipcc $CXXFLAGS -DFILLER_NK=$2 -DFILLER_BODY=$3 -include scripts/probe/filler_kernels.h -c cudecomp_amd/csrc/kernels_rows.hip -o $B/$1/kernels_rows.o || exit 1
      $ROCM/bin/hipcc $OBJS $B/$1/kernels_rows.o $LDFLAGS -o $B/$1/libcudecomp.so || exit 1
    else
      $ROCM/bin/hipcc $CXXFLAGS -DFILLER_NK=$2 -DFILLER_BODY=$3 -c scripts/probe/filler.hip -o $B/$1/filler.o || exit 1
      $ROCM/bin/hipcc $OBJS cudecomp_amd/build/kernels_rows.hip.o $B/$1/filler.o $LDFLAGS -o $B/$1/libcudecomp.so || exit 1
    fi
    ln -sf libcudecomp.so $B/$1/libcudecomp.so.0
    echo "$1: $(readelf -S -W $B/$1/libcudecomp.so | awk '/ .hip_fatbin /{print $6}') (hex) bytes of .hip_fatbin"
  }
  variant sep_few 26 256 0
  variant sep_many 118 16 0
  variant same_tu 80 256 1
  variant sep_small 9 256 0
  variant sep_mid 17 256 0
  # test program with its own +0.2 MB of device code
  $ROCM/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -DR32 -DFILLER_NK=26 -DFILLER_BODY=256 -include scripts/probe/filler_kernels.h -Iinclude \
      tests/native/transpose_test.cpp -Lcudecomp_amd/lib -lcudecomp -Wl,-rpath,$PWD/cudecomp_amd/lib -Wl,-rpath,$ROCM/lib -o $B/transpose_test_R32_padded || exit 1
  exit 0
fi
export HSA_ENABLE_IPC_MODE_LEGACY=0
python - <<'PY'
import os, re, sys, tempfile, time
ROOT = os.getcwd()
sys.path.insert(0, ROOT)
from tests.mp import run_binary_ranks
from tests.test_gpu_runner_cases import load_cases
lines = [l for l in load_cases()["transpose_test_cc"] if re.search(r"--backend [123678] ", l)][::3][:300]
exe = os.path.join(ROOT, "tests", "native", "build", "transpose_test_R32")
B = os.path.join(ROOT, "scripts", "probe", "bisect_libs")
def size(lib):
    import subprocess
    o = subprocess.run(["readelf", "-S", "-W", lib], capture_output=True, text=True).stdout
    return [int(l.split()[5], 16) for l in o.splitlines() if " .hip_fatbin " in l][0]
def run(label, libdir, env=None, exe_=exe, nranks=4):
    with tempfile.NamedTemporaryFile("w", suffix="_cases.txt", delete=False) as f:
        f.write("\n".join(lines) + "\n"); path = f.name
    e = dict(env or {})
    if libdir: e["LD_LIBRARY_PATH"] = libdir
    t0 = time.time()
    try:
        logs = run_binary_ranks(nranks, [exe_, "--testfile", path], timeout=600, extra_env=e)
        m = re.search(r"Completed all tests, running time ([0-9.]+) s", logs[0])
        ok = logs[0].count(" PASSED") == len(lines)
        ms = 1000 * float(m.group(1)) / len(lines) if m else -1
    except AssertionError as ex:
        ok, ms = False, -1; print(str(ex)[-500:])
    os.unlink(path)
    lib = os.path.join(libdir or os.path.join(ROOT, "cudecomp_amd", "lib"), "libcudecomp.so")
    print("%-72s .hip_fatbin %7d B  %6.1f ms per case  %s" % (label, size(lib), ms, "ok" if ok else "FAILED"), flush=True)
T = os.path.join(ROOT, "cudecomp_amd", "lib_tuning")
run("default library", None)
run("TUNING_VARIANTS library (real kernels)", T)
for v, what in (("sep_few", "default + separate code object, 26 large fillers"), ("sep_many", "default + separate code object, 118 small fillers"),
                ("same_tu", "default kernels + 26 large fillers in ONE code object"), ("sep_small", "default + separate code object (+0.07 MB)"),
                ("sep_mid", "default + separate code object (+0.13 MB)")):
    run(what, os.path.join(B, v))
run("TUNING_VARIANTS library, CUDECOMP_SKIP_LINK_PROBE=1", T, {"CUDECOMP_SKIP_LINK_PROBE": "1"})
run("TUNING_VARIANTS library, link probe with the copy kernel only", T, {"CUDECOMP_LINK_PROBE_ENGINES": "cu"})
run("TUNING_VARIANTS library, link probe with hipMemcpyAsync only", T, {"CUDECOMP_LINK_PROBE_ENGINES": "sdma"})
run("default library, test program with +0.2 MB of its own device code", None, exe_=os.path.join(B, "transpose_test_R32_padded"))
run("TUNING_VARIANTS library, 2 ranks", T, nranks=2)
run("default library again", None)
PY
