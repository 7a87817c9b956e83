#!/bin/bash
# N ranks of tests/c_abi/shard_client as N fresh processes (one per GPU, rank r on device r % n_devices), the RCCL unique id
# handed over through a file:   bash tests/c_abi/run_shards.sh <world> [n_devices]
# n_devices defaults to the GPUs rocm-smi lists (1 when it cannot tell).  Exit code: non-zero when any rank failed.
set -u
WORLD=${1:?world}
HERE=$(cd "$(dirname "$0")" && pwd)
EXE=$HERE/_build/shard_client
NDEV=${2:-$(ls -d /sys/class/kfd/kfd/topology/nodes/*/ 2>/dev/null | while read d; do [ "$(cat $d/simd_count 2>/dev/null || echo 0)" -gt 0 ] && echo $d; done | wc -l)}
[ "$NDEV" -ge 1 ] 2>/dev/null || NDEV=1
ID=$(mktemp -u /tmp/brie_comm_id.XXXXXX)
PIDS=()
for R in $(seq 0 $((WORLD - 1))); do
  # every rank under its own watchdog: a rank that dies before brie_comm_init leaves the others inside ncclCommInitRank
  HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 5 ${BRIE_SHARDS_TIMEOUT:-300} "$EXE" $R $WORLD $ID $NDEV &
  PIDS+=($!)
done
RC=0
for P in "${PIDS[@]}"; do wait $P || RC=1; done
rm -f $ID $ID.tmp
exit $RC
