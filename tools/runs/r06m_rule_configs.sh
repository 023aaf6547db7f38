#!/bin/bash
# after the attention rule change (no split form): attention + golden tests, and the other BASELINE / reference shapes end to end (configs[3], reference-default topology)
cd "$(dirname "$0")/../.."
O=gpurun_out/r06m; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -m gpu -q -x -k "attention or golden or full_size or config4 or bitwise" 2>&1 | tail -3 | tee $O/pytest_subset.txt
{ echo "# configs[3]: --variant ViT-L16 --batch 16 --patches 1024 --scales 3"; timeout 600 python3 tools/run_config.py --variant ViT-L16 --batch 16 --patches 1024 --scales 3 2>&1 | grep -v amdgpu
  echo "# reference default topology (train_config.py:169-194): --variant ViT-B16 --batch 16 --patches 512 --scales 5 --refdefault"; timeout 600 python3 tools/run_config.py --variant ViT-B16 --batch 16 --patches 512 --scales 5 --refdefault 2>&1 | grep -v amdgpu
  echo "# the same, one pair per forward"; timeout 600 python3 tools/run_config.py --variant ViT-B16 --batch 1 --patches 512 --scales 5 --refdefault 2>&1 | grep -v amdgpu; } > $O/configs.txt
cat $O/configs.txt
