cd /tmp; export TMPDIR=/tmp
for d in 0 63 56; do
  rm -rf /tmp/pa_$d
  CBIM_IGEMM_DBG=$d rocprofv3 --kernel-trace --stats -d /tmp/pa_$d -o p -- python $GRAFT_REPO_ROOT/tools/_abl.py > /dev/null 2>&1
  echo "== dbg $d"; python $GRAFT_REPO_ROOT/tools/rocpd_summary.py /tmp/pa_$d/p_results.db 1 2>&1 | head -12
done
