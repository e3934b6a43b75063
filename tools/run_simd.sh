cd $GRAFT_REPO_ROOT && ./tools/ubench_simd
