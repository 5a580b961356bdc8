#!/bin/bash
# Register / spill / scratch figures of every kernel of the library as the compiler reports them in the code-object
# metadata, read from the built objects (tools/obj_notes.sh; no recompilation) -- plus the instruction counts.
cd "$(dirname "$0")/../statusswitchingqp.jl_amd/csrc"
../../tools/obj_notes.sh ssqp_wave_v0.o ssqp_wave_v1.o ssqp_wave_v2.o ssqp_wave_lean_v0.o ssqp_wave_lean_v1.o ssqp_wave_lean_v2.o ssqp_full.o ssqp_kernels.o ssqp_phase1.o ssqp_phase1_wave.o
