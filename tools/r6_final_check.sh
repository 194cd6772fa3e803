# round 6: the whole -m gpu suite, smoke() and the default bench on the tree as it is (what the driver runs at round end)
O=gpurun_out/${1:-r6_final}
mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/suite.log 2>&1
echo "rc=$?" >> $O/suite.log
tail -3 $O/suite.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
cut -c1-250 $O/bench.json
