#!/bin/bash
# builds the microbenchmarks next to their sources (the binaries are git-ignored; they travel to the GPU box with gpurun)
cd "$(dirname "$0")"
for f in *.hip; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o "${f%.hip}" "$f" 2>&1 | grep -v warning; done
ls -la
