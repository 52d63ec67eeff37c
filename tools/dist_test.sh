#!/usr/bin/env bash
# Same command line as the reference's tools/dist_test.sh (CONFIG CHECKPOINT GPUS [test.py options ...]); test.py starts its own
# torch.distributed.run ranks (one per GPU, RCCL) when --launcher pytorch is given from a plain shell.
CONFIG=$1
CHECKPOINT=$2
GPUS=$3
exec python "$(dirname "$0")"/test.py "$CONFIG" "$CHECKPOINT" --launcher pytorch --gpus "$GPUS" "${@:4}"
