#!/bin/bash
# A/B of two builds of libge_step.so on the same GPU box, interleaved: tools/ab.sh a.so b.so ww:8:65536 [more shapes]
# (see abn.sh; variants are selected through GE_LIB_PATH, the product library is never overwritten)
A=$1; B=$2; shift 2
exec "$(dirname "$0")/abn.sh" "$*" "$A" "$B"
