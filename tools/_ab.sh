OUT=gpurun_out/ab1; mkdir -p $OUT
(timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $OUT/tests.log 2>&1
tail -5 $OUT/tests.log
for r in 0 1; do
  echo "== FS_RCP=$r headline"; FS_RCP=$r python tools/kbench.py --steps 20 --warm 30 --sweeps 100 2>&1 | grep -v "^#"
  echo "== FS_RCP=$r cfg1 (bc2 res1600)"; FS_RCP=$r python tools/kbench.py --res 1600 --bc 2 --steps 20 --warm 30 --sweeps 100 2>&1 | grep -v "^#"
  echo "== FS_RCP=$r cfg4 (bc3 kk)"; FS_RCP=$r python tools/kbench.py --bc 3 --scheme kk --vc 10 --steps 20 --warm 30 --sweeps 0 2>&1 | grep -v "^#"
done 2>&1 | tee $OUT/ab.log
