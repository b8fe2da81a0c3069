#!/bin/bash
L=$PWD/avir_amd/lib
g++ -O1 -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tests/cpp/window_guard.cpp -L$L -lavirhip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$L -Wl,-rpath,/opt/rocm/lib -o /tmp/window_guard || exit 1
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do /tmp/window_guard | grep -E "DIFF|differs|OK|fault" ; echo "run $i rc=$?"; done
