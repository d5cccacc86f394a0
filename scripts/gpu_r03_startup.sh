export HSA_ENABLE_IPC_MODE_LEGACY=0 CUDECOMP_PEER_TIMEOUT=30
cd $GRAFT_REPO_ROOT
cat > /tmp/one.txt <<'EOT'
--pr 2 --pc 4 --gx 32 --gy 30 --gz 34 --backend 6
--pr 2 --pc 4 --gx 32 --gy 30 --gz 34 --backend 6
EOT
for v in dev host; do
  for i in 1 2; do
  t0=$(date +%s.%N)
  for r in 0 1 2 3 4 5 6 7; do
    ( export RANK=$r WORLD_SIZE=8 LOCAL_RANK=$r MASTER_ADDR=127.0.0.1 MASTER_PORT=23456 CUDECOMP_BOOTSTRAP_PORT=2347$i CUDECOMP_VERBOSE=1 CUDECOMP_TEST_JOB=st$v$i
      [ $v = host ] && export CUDECOMP_FLAGS_IN_HOST_MEMORY=1
      tests/native/build/transpose_test_R64 --testfile /tmp/one.txt > /tmp/out_$v_$r.log 2>&1 ) &
  done
  wait
  t1=$(date +%s.%N)
  echo "$v flags run $i: $(echo "$t1 - $t0" | bc) s"; grep -h "CUDECOMP:" /tmp/out_$v_0.log | head -5
  done
done
