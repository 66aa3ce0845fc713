cd $GRAFT_REPO_ROOT
timeout 300 python tools/experiments/pageable_async_copy.py 2>&1 | tail -2
