for i in 1 2 3 4; do
  for sdma in 1 0; do
  for f in csvo esvo; do
    HSA_ENABLE_SDMA=$sdma python3 profiles/stream_bench.py --format $f --scene-depth 14 --radius 40 --width 3840 --height 2160 --frames 2 2>/dev/null | grep -o '"initial_fill": {[^}]*}' | head -1 | sed "s/^/sdma=$sdma $f /"
  done
  done
done
