# usage: bash tools/ab/sweep_lib.sh <lib> variant...   (tools/sweep.py under another build of the library)
lib=$1; shift
PT_LIBPTCORE=$PWD/$lib timeout -k 10 400 python3 tools/sweep.py --spp 64 --rounds 3 "$@" 2>&1 | grep -v amdgpu.ids | tail -12
