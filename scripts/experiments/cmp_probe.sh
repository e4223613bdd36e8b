cd $GRAFT_REPO_ROOT/tests/cpp/bin
for pol in 2 0; do
./cmp_probe.ref 2 $pol 1024 > /tmp/ref.txt
LBFGSX_PERSIST=0 ./cmp_probe.gpu 2 $pol 1024 > /tmp/gpu.txt
LBFGSX_PERSIST=0 ./cmp_probe.gpu 2 $pol 1024 > /tmp/gpu2.txt
echo "policy $pol: differing solves gpu vs ref: $(diff /tmp/ref.txt /tmp/gpu.txt | grep -c '^<'), gpu vs gpu: $(diff /tmp/gpu.txt /tmp/gpu2.txt | grep -c '^<')"
diff /tmp/ref.txt /tmp/gpu.txt | head -8
done
