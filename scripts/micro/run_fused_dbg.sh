# developer: stress and timing of the fused hypothesis decoder (default build)
python scripts/micro/fused_decoder_stress.py --reps 200 --views 8 2>&1 | grep -E "LDS_KB"
python scripts/micro/fused_decoder_stress.py --reps 100 --views 8 --noise 2>&1 | grep -E "LDS_KB"
python scripts/micro/fused_decoder_time.py 2>&1 | grep -E "LDS|max"
