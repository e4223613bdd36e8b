cd $GRAFT_REPO_ROOT/tests/cpp/bin
echo "== ref"; timeout 20 ./example-rosenbrock-comparison.ref | head -6
for v in "X=1" "X=1" "LBFGSX_PERSIST=0" "LBFGSX_FUSE_POST=0"; do echo "== gpu $v"; env $v timeout 25 ./example-rosenbrock-comparison.gpu | head -6; done
