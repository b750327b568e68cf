set -e
export TMPDIR=/tmp
O=gpurun_out/r3
mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/t_all3.log 2>&1 || { tail -60 $O/t_all3.log; exit 1; }
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
bash tools/r03_call11.sh > $O/call11b.log 2>&1
echo done
