#!/bin/bash
# the reference's render loop (one frame per call) with the canvas image asked for after every frame, against the
# same loop without it and against one batched call; scene_suzanne 1920x1080, 16 spp, 96 frames
cd "$(dirname "$0")/../.."
SC=tests/golden/scenes/scene_suzanne.json
for opt in "" "--per-frame" "--present-every 1" "--present-every 4"; do
  echo "cli.js $opt"
  timeout -k 10 120 node raytracer_amd/js/cli.js --scene $SC --width 1920 --height 1080 --spp 16 --frames 96 $opt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   ms/frame %.3f  Mrays/s %.0f' % (d['ms_per_frame'], d['mrays_per_s']))"
done
