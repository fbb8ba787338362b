#!/bin/bash
# free-running paths after the note-loop changes: the tests that touch them, then the tfr = 0 bench at B = 512 / 1024 and the decode
python -m pytest tests -m gpu -x -q -k "free or note_loop or decode or inference or sampling or golden or config" > gpurun_out/r06_free_tests.txt 2>&1
tail -4 gpurun_out/r06_free_tests.txt
for b in 512 1024; do
  timeout 300 python bench.py --tfr 0 --batch $b --no-cpu-baseline --no-parity --no-extras --steps 10 --warmup 3 2>&1 | grep '"metric"' | cut -c1-200
done
timeout 300 python bench.py --mode decode --batch 2048 --graph --no-cpu-baseline --no-parity --no-extras 2>&1 | grep '"metric"' | cut -c1-200
