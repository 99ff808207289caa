# round 6, call c: timeline of one lone proof (kernel trace, default build)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06c; mkdir -p $O
rm -rf /tmp/prof_t
rocprofv3 --kernel-trace -d /tmp/prof_t -o p -- python bench.py --no-cpu-baseline --no-h2d --no-verify --no-sharded-leg --quad-mix 0 --inflight 1 > $O/bench.json 2> $O/bench.err
db=$(find /tmp/prof_t -name "*.db" | head -1)
python -c "
import sqlite3; db=sqlite3.connect('$db'); print([r[1] for r in db.execute('pragma table_info(kernels)')])"
python tools/proof_timeline.py $db 10 > $O/timeline_proof10.md
python tools/proof_timeline.py $db 11 > $O/timeline_proof11.md
cat $O/timeline_proof10.md
