#!/bin/bash
# lane strides that are not powers of two: does the walkers' time depend on where their 64 private streams fall?
TUNING='{"lane_merge":1,"hp_dedupe":1}' bash tools/alone_trace.sh s_base | head -16
TUNING='{"lane_merge":1,"hp_dedupe":1,"hp_chunk":132096}' bash tools/alone_trace.sh s_hp | head -16
TUNING='{"lane_merge":1,"hp_dedupe":1,"hp_chunk":132096,"ar_chunk":33024,"mm_chunk":16640}' bash tools/alone_trace.sh s_all | head -16
