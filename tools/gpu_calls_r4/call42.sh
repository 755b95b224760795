cd $GRAFT_REPO_ROOT
echo "== two processes at once"
timeout 300 python tools/exp/adam_pack_stress.py 400 > /tmp/a.txt 2>&1 &
timeout 300 python tools/exp/adam_pack_stress.py 400 > /tmp/b.txt 2>&1 &
wait; tail -n 4 /tmp/a.txt; tail -n 4 /tmp/b.txt
