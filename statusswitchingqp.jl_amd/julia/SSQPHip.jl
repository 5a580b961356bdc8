# SSQPHip.jl -- Julia side of the drop-in boundary (UNVERIFIED: no Julia in the build image or on the GPU box).
#
# EXTENDS the reference's own generic function: after `using SSQPHip`, the more specific Float64 method below is
# what `StatusSwitchingQP.solveQP(Q, S, x0; settings)` dispatches to, so every caller of the reference reaches
# the GPU unchanged:
#
#   solveQP(Q::QP{Float64}, S, x0; settings)        replaces src/SSQP.jl:237-377  (the hot path)
#   solveQP(Q::QP{Float64}; settings, settingsLP)   src/SSQP.jl:224-234 keeps running the reference's own initQP
#                                                   (Phase-1) and then calls the 3-argument method (:233) -> GPU
#   MOI.optimize!(::Optimizer)                      src/MOIwrapper.jl:165 calls solveQP(opt.Problem; ...) -> GPU
#
# Everything else (QP/LP types, Settings, LP solvers, the MOI wrapper) stays the reference's.  BigFloat problems
# keep the reference's CPU method (the method below is for Float64 only).
#
# Usage:  ENV["SSQP_HIP_LIB"] = "/path/to/libssqp_hip.so"; using StatusSwitchingQP; include("SSQPHip.jl"); using .SSQPHip
#         z, S, status = solveQP(Q)        # the reference's entry point, same return triple
module SSQPHip

using LinearAlgebra
using StatusSwitchingQP
import StatusSwitchingQP: solveQP                     # methods are ADDED to the reference's function
using StatusSwitchingQP: QP, Settings, Status, DN

export solveQP, solveQP_batch, solveQP_full_hip, use_moi_qp_status!

const libssqp = get(ENV, "SSQP_HIP_LIB", "libssqp_hip.so")

# struct ssqp_settings { int32 maxIter; int32 rule; double tol; double tolG; }   (include/ssqp_hip.h)
struct CSettings
    maxIter::Int32
    rule::Int32
    tol::Float64
    tolG::Float64
end
function CSettings(s::Settings{Float64})
    s.rule == :Dantzig || error("SSQPHip: only rule=:Dantzig is implemented for Phase-1")
    CSettings(Int32(s.maxIter), Int32(0), s.tol, s.tolG)
end

# one context per GPU (ssqp_ctx_create(dev, ...)); created on first use
const _ctxs = Dict{Int,Ptr{Cvoid}}()
function ctx(dev::Integer=0)
    get!(_ctxs, Int(dev)) do
        r = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:ssqp_ctx_create, libssqp), Cint, (Cint, Ref{Ptr{Cvoid}}), dev, r)
        rc == 0 || error("ssqp_ctx_create($dev) failed with code $rc (2 = no HIP device; there is no CPU fallback)")
        r[]
    end
end

check(rc, c=ctx()) = rc == 0 || error("libssqp_hip: code $rc: " *
    unsafe_string(ccall((:ssqp_last_error, libssqp), Cstring, (Ptr{Cvoid},), c)))

# detail codes: 1/2 = a cholesky in the loop would have thrown, 3 = lu in Phase-1 would have thrown
rethrow_detail(detail) = detail in (1, 2) ? throw(LinearAlgebra.PosDefException(detail)) :
                         detail == 3 ? throw(LinearAlgebra.SingularException(0)) : nothing

"""
    solveQP(Q::QP{Float64}, S::Vector{Status}, x0::Vector{Float64}; settings=Settings{Float64}())

The hot path on the GPU (a method of `StatusSwitchingQP.solveQP`).  `S` is mutated in place and returned, `x0` is
not modified (SSQP.jl:264-266).  `throw_like_reference=true` rethrows the LAPACK exception the reference would have
raised where the library returns `status = -1` with a detail code.
"""
function solveQP(Q::QP{Float64}, S::Vector{Status}, x0::Vector{Float64}; settings=Settings{Float64}(),
                 throw_like_reference::Bool=false)
    length(S) == Q.N + Q.J || throw(DimensionMismatch("S must have length N+J"))
    z = Vector{Float64}(undef, Q.N)
    status = Ref{Int64}(0)
    detail = Ref{Int32}(0)
    cs = Ref(CSettings(settings))
    Si = reinterpret(Int32, S)                       # @enum Status is Int32: IN=0 DN=1 UP=2 OE=3 EO=4
    GC.@preserve Q S x0 z begin
        check(ccall((:ssqp_solve_f64, libssqp), Cint,
            (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
             Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ref{CSettings},
             Ref{Int64}, Ref{Int32}),
            ctx(), Q.N, Q.M, Q.J, Q.V, Q.A, Q.G, Q.q, Q.b, Q.g, Q.d, Q.u, Si, x0, z, cs, status, detail))
    end
    throw_like_reference && rethrow_detail(detail[])
    return z, S, Int(status[])
end

"""
    solveQP_full_hip(Q::QP{Float64}; settings, settingsLP)

`solveQP(Q)` with Phase-1 ALSO inside the library (its C++ restatement of initQP, SSQP.jl:461-560) instead of the
reference's Julia initQP.  Not installed as a method of `solveQP`: the reference's own one-argument method already
reaches the GPU through the three-argument method above.
"""
function solveQP_full_hip(Q::QP{Float64}; settings=Settings{Float64}(), settingsLP=settings,
                          throw_like_reference::Bool=false)
    if Q.mc <= 0
        return zeros(Float64, Q.N), fill(DN, Q.N), -1          # SSQP.jl:226-228
    end
    S = Vector{Status}(undef, Q.N + Q.J)
    z = Vector{Float64}(undef, Q.N)
    status = Ref{Int64}(0)
    detail = Ref{Int32}(0)
    cs = Ref(CSettings(settings))
    csl = Ref(CSettings(settingsLP))
    Si = reinterpret(Int32, S)
    GC.@preserve Q S z begin
        check(ccall((:ssqp_solve_full_f64, libssqp), Cint,
            (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
             Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cint, Ptr{Int32}, Ptr{Float64}, Ref{CSettings},
             Ref{CSettings}, Ref{Int64}, Ref{Int32}),
            ctx(), Q.N, Q.M, Q.J, Q.V, Q.A, Q.G, Q.q, Q.b, Q.g, Q.d, Q.u, Q.mc, Si, z, cs, csl, status, detail))
    end
    throw_like_reference && rethrow_detail(detail[])
    return z, S, Int(status[])
end

"""
    solveQP_batch(Qs::Vector{QP{Float64}}, Ss, x0s; settings, gpus=[0])

A batch of equal-shape QPs in ONE call, cut into contiguous blocks over the listed GPUs
(`ssqp_solve_batch_multi_f64`: one context and one host thread per GPU, no exchange between the blocks; the
results of every block land in the arrays returned here).  `Ss[p]` is mutated in place like in `solveQP`.
"""
function solveQP_batch(Qs::Vector{QP{Float64}}, Ss::Vector{Vector{Status}}, x0s::Vector{Vector{Float64}};
                       settings=Settings{Float64}(), gpus=[0])
    P = length(Qs)
    P == 0 && return Vector{Float64}[], Ss, Int[]
    N, M, J = Qs[1].N, Qs[1].M, Qs[1].J
    all(q -> (q.N, q.M, q.J) == (N, M, J), Qs) || throw(DimensionMismatch("solveQP_batch needs equal shapes"))
    pack(f, len) = (a = Vector{Float64}(undef, len * P); for p in 1:P; copyto!(a, (p - 1) * len + 1, vec(f(Qs[p])), 1, len); end; a)
    V = pack(q -> q.V, N * N); A = pack(q -> q.A, M * N); G = pack(q -> q.G, J * N)
    qv = pack(q -> q.q, N); b = pack(q -> q.b, M); g = pack(q -> q.g, J); d = pack(q -> q.d, N); u = pack(q -> q.u, N)
    S = Vector{Int32}(undef, (N + J) * P)
    x0 = Vector{Float64}(undef, N * P)
    for p in 1:P
        copyto!(S, (p - 1) * (N + J) + 1, reinterpret(Int32, Ss[p]), 1, N + J)
        copyto!(x0, (p - 1) * N + 1, x0s[p], 1, N)
    end
    z = Vector{Float64}(undef, N * P)
    status = Vector{Int64}(undef, P)
    detail = Vector{Int32}(undef, P)
    cs = Ref(CSettings(settings))
    cx = [ctx(dv) for dv in gpus]
    check(ccall((:ssqp_solve_batch_multi_f64, libssqp), Cint,
        (Ptr{Ptr{Cvoid}}, Cint, Cint, Cint, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
         Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ptr{Float64}, Ptr{Float64},
         Ref{CSettings}, Ptr{Int64}, Ptr{Int32}, Ptr{Cvoid}),
        cx, length(cx), P, N, M, J, V, A, G, qv, b, g, d, u, S, x0, z, cs, status, detail, C_NULL), cx[1])
    for p in 1:P
        copyto!(reinterpret(Int32, Ss[p]), 1, S, (p - 1) * (N + J) + 1, N + J)
    end
    return [z[(p - 1) * N + 1:p * N] for p in 1:P], Ss, Int.(status)
end

"""
    solveQP_multipliers(Q::QP{Float64}, S, x0; settings) -> (z, S, status, lambda, gamma)

`solveQP(Q, S, x0)` that also returns the Lagrange multipliers of the last pass, which the reference computes inside the
loop (`alphaL`, `gamma`, SSQP.jl:351-352; purged rows SSQP.jl:158-159) and never returns: `lambda` by row of `[A; G]`
(length M+J), `gamma` by variable (length N, 0 on the free ones).  One `ccall` of `ssqp_solve_batch_f64` with one problem.
"""
function solveQP_multipliers(Q::QP{Float64}, S::Vector{Status}, x0::Vector{Float64}; settings=Settings{Float64}())
    z = Vector{Float64}(undef, Q.N); status = Ref{Int64}(0); detail = Ref{Int32}(0)
    lambda = zeros(Q.M + Q.J); gamma = zeros(Q.N)
    cs = Ref(CSettings(settings.maxIter, 0, settings.tol, settings.tolG))
    Si = reinterpret(Int32, S)
    GC.@preserve S begin
        check(ccall((:ssqp_solve_batch_f64, libssqp), Cint,
            (Ptr{Cvoid}, Cint, Cint, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
             Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ref{CSettings},
             Ref{Int64}, Ref{Int32}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}),
            ctx(), 1, Q.N, Q.M, Q.J, Q.V, Q.A, Q.G, Q.q, Q.b, Q.g, Q.d, Q.u, Si, x0, z, cs, status, detail, C_NULL,
            lambda, gamma))
    end
    return z, S, Int(status[]), lambda, gamma
end

"""
    use_moi_qp_status!()

DELIBERATE DEVIATION, opt-in.  `MOI.get(::Optimizer, ::MOI.TerminationStatus)` of the reference maps `Results[3]`
with the LP code table (MOIwrapper.jl:213-228), but for a QP `Results[3]` is the loop's pass count
(MOIwrapper.jl:165, SSQP.jl:374): a QP that converges in 3 passes is reported INFEASIBLE_OR_UNBOUNDED, in 4 or more
ITERATION_LIMIT.  This installs a method that reports `status > 0` as OPTIMAL when the stored problem is a QP and
otherwise defers to the reference's table.
"""
function use_moi_qp_status!()
    @eval begin
        import MathOptInterface as MOI
        function MOI.get(opt::StatusSwitchingQP.Optimizer, ::MOI.TerminationStatus)
            opt.Results === nothing && return MOI.OPTIMIZE_NOT_CALLED
            st = opt.Results[3]
            if opt.Problem isa StatusSwitchingQP.QP
                return st > 0 ? MOI.OPTIMAL : st == 0 ? MOI.INFEASIBLE : st == -1 ? MOI.NUMERICAL_ERROR : MOI.ITERATION_LIMIT
            end
            return st == 3 ? MOI.INFEASIBLE_OR_UNBOUNDED : st in (1, 2) ? MOI.OPTIMAL : st == 0 ? MOI.INFEASIBLE :
                   st == -1 ? MOI.NUMERICAL_ERROR : MOI.ITERATION_LIMIT
        end
    end
    nothing
end

end # module
