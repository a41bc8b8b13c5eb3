cd $GRAFT_REPO_ROOT
bash tools/ab_time.sh r04r2
