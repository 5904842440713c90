for g in 2 3 4 6 8; do TAG=g$g bash scripts/gpu_joint_once.sh --group $g --batches $(( g >= 6 ? 6 : 12 )); done
TAG=g4r1 bash scripts/gpu_joint_once.sh --group 4 --replicas 1
TAG=g8r1 bash scripts/gpu_joint_once.sh --group 8 --replicas 1 --batches 6
