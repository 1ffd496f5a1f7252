#!/bin/bash
cd /root/repo
timeout 300 python tools/lba_probe.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl"
