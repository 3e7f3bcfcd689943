#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04big; mkdir -p $O
( timeout 2400 python tools/order_quality.py --users 1000000 --items 200000 --seeds 1 --oracle-variants 0 --engine-variants "workgroups=128" ) > $O/quality_1m_200k.log 2>&1; grep -v amdgpu.ids $O/quality_1m_200k.log | tail -6
