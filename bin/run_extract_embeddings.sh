#!/bin/bash
# Counterpart of egs/voxceleb/v1/nnet/run_extract_embeddings.sh (same options, positional arguments, stages and
# output files: xvector.JOB.{ark,scp}, xvector.scp, spk_xvector.{ark,scp}, num_utts.ark, log/extract.JOB.log),
# without Kaldi binaries: one fresh process per GPU, utterances sharded by frame count.
#   bin/run_extract_embeddings.sh [--nj N] [--min-chunk-size 50] [--chunk-size 10000] [--stage 0]
#        [--normalize false] [--checkpoint -1] [--node output] <nnet-dir> <data> <embeddings-dir>
echo "$0 $@"
if [ -f path.sh ]; then . ./path.sh; fi
here=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
export PYTHONPATH=$here:$PYTHONPATH
if [ $# -lt 3 ]; then
  echo "Usage: $0 [options] <nnet-dir> <data> <embeddings-dir>"
  echo "Options:"
  echo "  --use-gpu <true>"
  echo "  --nj <number of visible GPUs>"
  echo "  --min-chunk-size <50>"
  echo "  --chunk-size <10000>"
  echo "  --normalize <false>"
  echo "  --checkpoint <-1>"
  echo "  --node <output>"
  echo ""
  exit 100
fi
# python, not exec: the launcher starts its jobs as child processes and never touches a GPU before they end
python -m tf_kaldi_speaker_amd.run_extract "$@"
exit $?
