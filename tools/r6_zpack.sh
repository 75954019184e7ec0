cd $GRAFT_REPO_ROOT
for i in 1 2; do
echo "== base"; python tools/fwd_h2_check.py 2>&1 | grep " h2 "
echo "== zpack (timing variant, wrong results)"; YT8M_LIB=$PWD/tools/variants/lib_zpack.so python tools/fwd_h2_check.py 2>&1 | grep " h2 "
done
