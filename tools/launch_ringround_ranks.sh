#!/bin/bash
# One process per GPU without Python: starts N copies of examples/ringround_multi (rank r on device r), which meet through RCCL
# (alch_comm_unique_id -> a file -> alch_comm_init_rank; include/alchemy_rccl.h).  Prints every rank's JSON line; exit status 0 only
# when every rank's checks passed.   usage: tools/launch_ringround_ranks.sh N [--batch B] [--passes K] [--gather G] [--lanes S]
set -u
N=${1:-1}; shift || true
root="$(cd "$(dirname "$0")/.." && pwd)"
cd "$root"
[ -x examples/ringround_multi ] || tools/build_examples.sh
dir=$(mktemp -d /tmp/alch_ranks.XXXXXX)
pids=()
for r in $(seq 0 $((N - 1))); do
    ./examples/ringround_multi --world "$N" --rank "$r" --device "$r" --id-file "$dir/id" "$@" > "$dir/rank_$r.json" 2> "$dir/rank_$r.err" &
    pids+=($!)
done
rc=0
# a rank that fails before or inside a collective would leave its peers waiting in RCCL for ever: the first failure ends them all
left=$N
while [ "$left" -gt 0 ]; do
    if wait -n; then :; else
        rc=1
        for p in "${pids[@]}"; do kill "$p" 2> /dev/null; done
    fi
    left=$((left - 1))
done
for r in $(seq 0 $((N - 1))); do
    cat "$dir/rank_$r.json"
    [ -s "$dir/rank_$r.err" ] && sed "s/^/rank $r: /" "$dir/rank_$r.err" >&2
done
rm -rf "$dir"
exit $rc
