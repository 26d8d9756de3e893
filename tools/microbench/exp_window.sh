for cfg in "256 128 0" "512 64 0" "1024 32 0" "512 128 0" "1024 64 0" "1024 128 0" "256 128 1" "256 256 1" "256 1024 1" "1024 64 1"; do
  set -- $cfg
  echo "threads/block=$1 blocks=$2 blocked=$3 -> waves=$(( $1 / 64 * $2 ))"
  CW_TUNE_RENDER_BLOCKED=$3 CW_TUNE_RENDER_THREADS=$1 CW_TUNE_RENDER_BLOCKS=$2 CW_TUNE_RENDER_BLOCKS_PER_CU=8 python tools/microbench/time_render.py preceding 2>/dev/null | head -1
done
