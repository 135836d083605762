#!/bin/bash
# round 3, GPU session 30: packed-f32 VALU chains beside another kernel's MFMAs (tools/ubench/pk_beside_mfma.hip)
set -u
timeout 300 ./tools/ubench/pk_beside_mfma
