#!/bin/bash
export TESTS="-k 'headline_shape_stepwise or gist_stepwise'"
export CONFIGS="ab_libs/chain3.so;ab_libs/chain2.so;ab_libs/chain3.so COGAPS_NO_CHAIN=1" TAG=chain3
bash tools/r4_chain_ab.sh
