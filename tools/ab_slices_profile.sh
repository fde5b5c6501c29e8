for v in -1 512 128 64; do
  HCSPMM_SLICE_THRESHOLD=$v PROF_OUT=gpurun_out/prof_slices/thr_$v SKIP_MFMA=1 bash tools/profile_round.sh reddit_d128 || exit 1
done
