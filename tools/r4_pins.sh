#!/bin/bash
# regenerate the GPU iteration pins, then run the whole GPU suite with them
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python tools/make_gpu_iteration_pins.py gpurun_out/gpu_iteration_counts.json > gpurun_out/pins.log 2>&1
cp gpurun_out/gpu_iteration_counts.json tests/golden/gpu_iteration_counts.json
bash tools/gpu_pytest.sh r4_full_b tests -m gpu -q
