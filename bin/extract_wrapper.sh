#!/bin/bash
# Counterpart of egs/voxceleb/v1/nnet/wrap/extract_wrapper.sh (same options and positional
# arguments); runs this repository's extraction driver on one MI355X.

env=
gpuid=-1
min_chunk_size=25
chunk_size=10000
normalize=false
node="output"

if [ -f path.sh ]; then . ./path.sh; fi
if [ -f parse_options.sh ] || command -v parse_options.sh >/dev/null 2>&1; then
  . parse_options.sh || exit 1;
else
  # minimal --name value parser when Kaldi's utils/parse_options.sh is not on PATH
  while [ $# -gt 0 ]; do
    case "$1" in
      --*) name=$(echo "${1#--}" | tr '-' '_'); eval "$name=\"$2\""; shift 2 ;;
      *) break ;;
    esac
  done
fi

if [ $# != 3 ]; then
  echo "Usage: $0 [options] <nnet-dir> <data> <embeddings-dir>"
  echo "Options:"
  echo "  --gpuid <-1>"
  echo "  --min-chunk-size <25>"
  echo "  --chunk-size <10000>"
  echo "  --normalize <false>"
  echo "  --node <output>"
  echo ""
  exit 100
fi

nnetdir=$1
feat=$2
dir=$3

cmdopt_norm=
if $normalize; then
  cmdopt_norm="--normalize"
fi

here=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
export PYTHONPATH=$here:$PYTHONPATH

python -m tf_kaldi_speaker_amd.extract --gpu $gpuid --node $node --min-chunk-size $min_chunk_size \
       --chunk-size $chunk_size $cmdopt_norm "$nnetdir" "$feat" "$dir"
