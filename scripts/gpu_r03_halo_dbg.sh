#!/bin/bash
mkdir -p gpurun_out/halodbg
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/halodbg
T="tests/test_gpu_baseline_configs.py::test_config5_halo_full_size"
( CUDECOMP_VERBOSE=1 CUDECOMP_PEER_TIMEOUT=30 timeout 400 python -m pytest "$T" -x -q -k mpi_xyz 2>&1 | tail -15 | cut -c1-300 ) > $O/a_default.log 2>&1; tail -3 $O/a_default.log
( CUDECOMP_FLAGS_IN_HOST_MEMORY=1 CUDECOMP_PEER_TIMEOUT=30 timeout 400 python -m pytest "$T" -x -q -k mpi_xyz 2>&1 | tail -15 | cut -c1-300 ) > $O/b_hostflags.log 2>&1; tail -3 $O/b_hostflags.log
( CUDECOMP_PEER_COPY_ENGINE=sdma CUDECOMP_PEER_TIMEOUT=30 timeout 400 python -m pytest "$T" -x -q -k mpi_xyz 2>&1 | tail -15 | cut -c1-300 ) > $O/c_sdma.log 2>&1; tail -3 $O/c_sdma.log
( CUDECOMP_PEER_TIMEOUT=30 timeout 300 python -m pytest tests/test_gpu_self_exchange.py -x -q 2>&1 | tail -5 | cut -c1-300 ) > $O/d_self.log 2>&1; tail -3 $O/d_self.log
