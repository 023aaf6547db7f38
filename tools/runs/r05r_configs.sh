#!/bin/bash
# the other BASELINE / reference shapes end to end on the round-5 tree
O=gpurun_out/r05r; mkdir -p $O
{ echo "# configs[3]: --variant ViT-L16 --batch 16 --patches 1024 --scales 3"; timeout 600 python3 tools/run_config.py --variant ViT-L16 --batch 16 --patches 1024 --scales 3 2>&1 | grep -v amdgpu
  echo "# configs[3]'s model, one pair per forward: --variant ViT-L16 --batch 1 --patches 1024 --scales 3"; timeout 600 python3 tools/run_config.py --variant ViT-L16 --batch 1 --patches 1024 --scales 3 2>&1 | grep -v amdgpu
  echo "# reference default topology (train_config.py:169-194): --variant ViT-B16 --batch 16 --patches 512 --scales 5 --refdefault"; timeout 600 python3 tools/run_config.py --variant ViT-B16 --batch 16 --patches 512 --scales 5 --refdefault 2>&1 | grep -v amdgpu
  echo "# the same, one pair per forward"; timeout 600 python3 tools/run_config.py --variant ViT-B16 --batch 1 --patches 512 --scales 5 --refdefault 2>&1 | grep -v amdgpu; } > $O/configs.txt
cat $O/configs.txt
