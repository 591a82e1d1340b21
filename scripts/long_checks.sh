#!/bin/bash
# The long forms of the randomized GPU tests (run on the GPU box from the repo root; ~4 minutes):
#   decoder: 20000 random files (oracle encoder + libjpeg-turbo through Pillow, restart intervals, optimised tables) against libjpeg,
#            then 6000 more each also as a damaged copy (statuses only, never a fault)
#   encoder: 8000 random images against the CPU restatement of libjpeg
# Round 4, final build (placement pools, LUT index rounding): 1000 pixel-path configurations -- two in three also through the LUT pipelines --, 20 000 + 6000 files through the decoder, 8000 images through the encoder: 0 mismatches.  Earlier in the round: 2000 pixel-path configurations after the exact-fma front end of generate, 0 mismatches.  Round 3: 1500, 0 mismatches.  Last full run (round 2, final build): 1000 pixel-path configurations, 40 000 + 12 000 identical decodes, 0 mismatches; 8000 identical encodes.
set -e
# pixel path: 1000 random configurations (sizes, strides, gamuts, transfer functions, map scales, output formats, display boosts,
# EXACT on every fourth) against the oracle
UHDR_FUZZ_SEEDS=1000 python -m pytest tests/test_gpu_fuzz.py -x -q
python tests/stress_jpeg_dec.py 20000 11
python tests/stress_jpeg_dec.py 6000 23 damage
UHDR_ENC_SWEEP=8000 UHDR_ENC_SWEEP_SEED=5 python -m pytest tests/test_gpu_jpeg.py -x -q -k random_images
