cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s2h
mkdir -p $O
MIOPEN_LOG_LEVEL=5 timeout -k 10 500 python tools/wino_probe.py 256 > $O/probe.txt 2> $O/probe_log.txt
cat $O/probe.txt
grep -o "Solver [A-Za-z0-9_]*\|solver_id = [A-Za-z0-9_]*\|[A-Za-z]*Winograd[A-Za-z0-9_]*" $O/probe_log.txt | sort | uniq -c | sort -rn | head -30
