cd $GRAFT_REPO_ROOT
for zc in 1 0; do
  echo "== JXL_COMMIT_ZEROCOPY=$zc"
  for n in 1 2 4 8 12; do JXL_COMMIT_ZEROCOPY=$zc python3 tools/r4_stream.py $n 8 2>&1 | tail -2; done
done
