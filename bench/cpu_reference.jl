# cpu_reference.jl -- times the REFERENCE (StatusSwitchingQP.jl, Julia + OpenBLAS) on bench.py's workload.
# UNVERIFIED: neither the build image nor the GPU box has Julia; BASELINE.md's Julia column stays blank until
# someone with Julia runs this.  The inputs are bit-identical to bench.py's: they come from the library's own
# generator through ccall (ssqp_generate_batch, include/ssqp_hip.h; it needs no GPU).
#
#   julia -t auto bench/cpu_reference.jl /path/to/libssqp_hip.so [cfg4] [nprob=64]
#
# One QP per Julia thread, BLAS.set_num_threads(1) (the reference has no parallelism of its own); reports
# QPs/s of solveQP(Q, S, x0) (SSQP.jl:237 -- the hot path, Phase-1 outside the timed region) and of solveQP(Q).
using LinearAlgebra, StatusSwitchingQP, Printf
const lib = ARGS[1]
const cfgname = length(ARGS) > 1 ? ARGS[2] : "cfg4"
const P = length(ARGS) > 2 ? parse(Int, ARGS[3]) : 64
struct GenCfg; N::Int32; M::Int32; J::Int32; T::Int32; delta::Float64; ub::Float64; gscale::Float64; qscale::Float64; end
const CFG = Dict("cfg1" => GenCfg(50, 1, 0, 100, 1e-3, 0.0, 1.2, 0.0), "cfg2" => GenCfg(512, 1, 10, 1024, 1e-3, 3 / 64, 1.2, 0.1),
                 "cfg3" => GenCfg(256, 1, 0, 512, 1e-3, 3 / 32, 1.2, 0.0), "cfg4" => GenCfg(512, 1, 10, 1024, 1e-3, 3 / 64, 1.2, 0.1))
c = CFG[cfgname]; N, M, J = Int(c.N), Int(c.M), Int(c.J)
V = zeros(N * N * P); A = zeros(M * N * P); G = zeros(J * N * P); q = zeros(N * P); b = zeros(M * P); g = zeros(J * P)
d = zeros(N * P); u = zeros(N * P)
rc = ccall((:ssqp_generate_batch, lib), Cint, (Ref{GenCfg}, UInt64, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
           Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cint), c, 20261003, P, V, A, G, q, b, g, d, u, 0)
rc == 0 || error("generator failed: $rc")
sl(a, len, p) = a[(p - 1) * len + 1:p * len]
Qs = [QP(reshape(sl(V, N * N, p), N, N); q=sl(q, N, p), A=reshape(sl(A, M * N, p), M, N), b=sl(b, M, p),
         G=reshape(sl(G, J * N, p), J, N), g=sl(g, J, p), d=sl(d, N, p), u=sl(u, N, p)) for p in 1:P]
BLAS.set_num_threads(1)
starts = [StatusSwitchingQP.SSQP.initQP(Q, Settings{Float64}()) for Q in Qs]      # (x0, S, status), SSQP.jl:461
solveQP(Qs[1], copy(starts[1][2]), starts[1][1])                                    # compile
t = @elapsed Threads.@threads for p in 1:P
    solveQP(Qs[p], copy(starts[p][2]), starts[p][1])
end
@printf("%s: %d QPs, %d threads: solveQP(Q,S,x0) %.1f QPs/s\n", cfgname, P, Threads.nthreads(), P / t)
t = @elapsed Threads.@threads for p in 1:P
    solveQP(Qs[p])
end
@printf("%s: %d QPs, %d threads: solveQP(Q)      %.1f QPs/s (Phase-1 included)\n", cfgname, P, Threads.nthreads(), P / t)
