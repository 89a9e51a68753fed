# SimulatedAnnealingABCHIP.jl -- thin Julia binding of libsabc_hip.so (include/sabc_hip.h).
#
# Keeps the reference's public surface for the hot path -- `sabc`, `update_population!`,
# `SABCresult`, `SABCstate`, `RandomWalk`, `DifferentialEvolution`, `StretchMove`
# (src/SimulatedAnnealingABC.jl:19,28-60,251-259,451-460; src/proposals.jl:6) -- and does
# nothing but marshal arguments into `ccall`s.  All numerics live in the library.
#
# NOT EXECUTED IN THE BUILD CONTAINER: Julia is not installed there (SURVEY.md section 8c).  The
# Python mirror in ../api.py makes exactly the same calls and is what the test-suite drives; what CAN be
# checked without Julia is checked by tests/test_julia_binding.py: the field order, widths and offsets of
# CConfig / CUpdateArgs against sizeof / offsetof of the C structs, every ccall's symbol, return type and
# argument list against the prototypes of include/sabc_hip.h, and the enum values used below.
module SimulatedAnnealingABCHIP

using Distributions: Distribution, Normal, Uniform, Exponential, LogNormal, Gamma, Beta, Truncated, MvNormal, UnivariateDistribution,
                     ContinuousMultivariateDistribution, logpdf
using LinearAlgebra: cholesky, Symmetric
using ProgressMeter: Progress, next!, finish!          # same progress UI as the reference (:290-292,374)
import Dates
import Base: show

export sabc, update_population!, RandomWalk, DifferentialEvolution, StretchMove,
       DeviceDistance, GaussianIID, Gaussian2D, GandK, LotkaVolterra, DeviceSource, SourcePrior, comm_unique_id

const libsabc = get(ENV, "SABC_HIP_LIB", joinpath(@__DIR__, "..", "libsabc_hip.so"))

const MAX_PARA, MAX_STATS, MAX_MODEL_PARAMS = 16, 64, 32
const MAX_PARA2 = MAX_PARA * MAX_PARA      # the row-major Cholesky factor of an MvNormal prior

# ---- C structs (must match include/sabc_hip.h field for field) ----
struct CConfig
    abi_version::Int32
    device::Int32
    n_particles::Int64
    n_para::Int32
    n_stats::Int32
    model_id::Int32
    n_model_params::Int32
    model_params::NTuple{MAX_MODEL_PARAMS,Float64}
    prior_kind::NTuple{MAX_PARA,Int32}
    prior_a::NTuple{MAX_PARA,Float64}
    prior_b::NTuple{MAX_PARA,Float64}
    prior_c::NTuple{MAX_PARA,Float64}
    prior_d::NTuple{MAX_PARA,Float64}
    prior_joint::Int32
    reserved2::Int32
    prior_chol::NTuple{MAX_PARA2,Float64}
    algorithm::Int32
    rank::Int32
    world::Int32
    reserved::Int32
    v::Float64
    delta::Float64
    seed::UInt64
end

struct CUpdateArgs
    n_simulation::Int64
    v::Float64
    delta::Float64
    resample::Float64
    checkpoint_history::Int64
    proposal_kind::Int32
    more_chunks_follow::Int32
    proposal_p0::Float64
    proposal_p1::Float64
    history_phase::Int64
end

# ---- proposals: same constructors and errors as src/proposals.jl ----
abstract type Proposal end

mutable struct RandomWalk{T} <: Proposal
    β::Float64
    Σ::T
end
function RandomWalk(; β=0.8, n_para)
    (0 < β <= 1) || error("Mixing parameter `β` must be between zero and one.")   # proposals.jl:30
    n_para == 1 ? RandomWalk(β, -1.0) : RandomWalk(β, -ones(n_para, n_para))
end

struct DifferentialEvolution <: Proposal
    γ0::Float64
    σ_gamma::Float64
    function DifferentialEvolution(; γ0=nothing, n_para=nothing, σ_gamma=1e-5)
        if !isnothing(γ0) && isnothing(n_para)
            new(γ0, σ_gamma)
        elseif !isnothing(n_para) && isnothing(γ0)
            new(2.38 / sqrt(2 * n_para), σ_gamma)                                  # proposals.jl:93
        else
            throw(ArgumentError("Provide either `γ0` or `n_para`, not both."))   # proposals.jl:96
        end
    end
end

struct StretchMove <: Proposal
    a::Float64
end
StretchMove(; a=2) = StretchMove(a)

descriptor(p::RandomWalk) = (Int32(0), p.β, 0.0)
descriptor(p::DifferentialEvolution) = (Int32(1), p.γ0, p.σ_gamma)
descriptor(p::StretchMove) = (Int32(2), p.a, 0.0)

# ---- device-coded simulators: `f_dist` as data (<: Function so that sabc(f_dist::Function, ...) dispatches) ----
abstract type DeviceDistance <: Function end
struct GaussianIID <: DeviceDistance
    n_obs::Int; sd::Float64; obs_mean::Float64; obs_m2::Union{Nothing,Float64}
end
GaussianIID(; n_obs=100, sd=1.0, obs_mean=0.0, obs_m2=nothing) = GaussianIID(n_obs, sd, obs_mean, obs_m2)
struct Gaussian2D <: DeviceDistance
    n_obs::Int; r::Float64; obs_mean::NTuple{2,Float64}; obs_varsum::Float64; obs_cov::Float64
end
struct GandK <: DeviceDistance
    n_draws::Int; c::Float64; ranks::NTuple{4,Int}; obs::NTuple{4,Float64}
end
struct LotkaVolterra <: DeviceDistance
    n_steps::Int; dt::Float64; σ::Float64; x0::Float64; y0::Float64; obs::NTuple{4,Float64}
end
model_id(::GaussianIID) = Int32(1); model_id(::Gaussian2D) = Int32(2)
model_id(::GandK) = Int32(3); model_id(::LotkaVolterra) = Int32(4)
n_stats(m::GaussianIID) = isnothing(m.obs_m2) ? 1 : 2
n_stats(::Gaussian2D) = 3; n_stats(::GandK) = 4; n_stats(::LotkaVolterra) = 4
params(m::GaussianIID) = Float64[m.n_obs, m.sd, m.obs_mean, something(m.obs_m2, 0.0)]
params(m::Gaussian2D) = Float64[m.n_obs, m.r, m.obs_mean..., m.obs_varsum, m.obs_cov]
params(m::GandK) = Float64[m.n_draws, m.c, m.ranks..., m.obs...]
params(m::LotkaVolterra) = Float64[m.n_steps, m.dt, m.σ, m.x0, m.y0, m.obs...]

# ---- any other f_dist: stays a Julia function, called back by the library for the proposals that
#      passed the prior gate (include/sabc_hip.h: sabc_simulate_fn); everything else runs on the GPU
struct HostDistance{F} <: DeviceDistance
    f::F
    n_stats::Int
    n_para::Int
    args::Tuple
    kwargs::NamedTuple
end
model_id(::HostDistance) = Int32(0)

# ---- f_dist as HIP source, compiled at run time into the fused update kernel (include/sabc_hip.h:
#      sabc_register_device_simulator); with a SourcePrior the same source also defines the prior (prior_joint = 3) ----
struct DeviceSource <: DeviceDistance
    source::String
    n_para::Int
    n_stats::Int
    params::Vector{Float64}
end
DeviceSource(source, n_para, n_stats; params=Float64[]) = DeviceSource(source, n_para, n_stats, collect(Float64, params))
model_id(::DeviceSource) = Int32(5)
n_stats(m::DeviceSource) = m.n_stats
params(m::DeviceSource) = m.params
"""
    SourcePrior(d)

ANY prior of `d` parameters next to a `DeviceSource`: its HIP source defines `sabc_user_prior_sample` / `sabc_user_prior_logpdf`
(include/sabc_hip.h), and rand / logpdf of SimulatedAnnealingABC.jl:174,314,318 run inside the fused kernel.
"""
struct SourcePrior <: ContinuousMultivariateDistribution
    d::Int
end
Base.length(p::SourcePrior) = p.d
n_stats(m::HostDistance) = m.n_stats
params(::HostDistance) = Float64[]

# `@cfunction($cb, ...)` builds a runtime closure: supported on x86-64 / aarch64 Linux (where MI355X hosts run), not on
# every platform Julia supports.  The returned Base.CFunction must stay rooted for as long as the library may call it:
# it is kept in HOST_CALLBACKS until the handle's finalizer has run.
function host_callback(m::HostDistance)
    function cb(ctx::Ptr{Cvoid}, theta::Ptr{Float64}, ids::Ptr{Int64}, n::Int64, iter::UInt64, rho::Ptr{Float64})::Cint
        try
            Θ = unsafe_wrap(Array, theta, (Int(n), m.n_para))        # column-major n x d
            R = unsafe_wrap(Array, rho, (Int(n), m.n_stats))
            Threads.@threads for i in 1:n                             # like SimulatedAnnealingABC.jl:308
                θ = m.n_para == 1 ? Θ[i, 1] : Θ[i, :]
                R[i, :] .= collect(Float64, m.f(θ, m.args...; m.kwargs...))
            end
            return Cint(0)
        catch
            return Cint(-1)
        end
    end
    @cfunction($cb, Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Int64}, Int64, UInt64, Ptr{Float64}))
end

# ---- priors as data: (kind, a, b, c, d) per dimension, include/sabc_hip.h SABC_PRIOR_* ----
prior_descriptor(d::Normal) = (Int32(0), d.μ, d.σ, 0.0, 0.0)
prior_descriptor(d::Uniform) = (Int32(1), d.a, d.b, 0.0, 0.0)
prior_descriptor(d::Exponential) = (Int32(2), d.θ, 0.0, 0.0, 0.0)
prior_descriptor(d::LogNormal) = (Int32(3), d.μ, d.σ, 0.0, 0.0)
prior_descriptor(d::Gamma) = (Int32(4), d.α, d.θ, 0.0, 0.0)
prior_descriptor(d::Beta) = (Int32(5), d.α, d.β, 0.0, 0.0)
prior_descriptor(d::Truncated{<:Normal}) = (Int32(6), d.untruncated.μ, d.untruncated.σ, Float64(d.lower), Float64(d.upper))
prior_descriptors(d::UnivariateDistribution) = [prior_descriptor(d)]
# MvNormal(mu, Sigma): mu travels in the per-dimension descriptors, the lower Cholesky factor of Sigma in prior_chol
prior_descriptors(d::MvNormal) = [(Int32(0), d.μ[k], sqrt(d.Σ[k, k]), 0.0, 0.0) for k in 1:length(d)]
prior_chol(d::Distribution) = (Int32(0), Float64[])
function prior_chol(d::MvNormal)
    n = length(d)
    L = cholesky(Symmetric(Matrix(d.Σ))).L
    (Int32(1), Float64[l <= k ? L[k, l] : 0.0 for k in 1:n for l in 1:n])        # row-major n x n
end
# product_distribution([...]): Distributions.jl names the vector of marginals `v` (Product, <= 0.25.x) or `dists`
# (ProductDistribution); neither has an exported accessor, so both spellings are accepted and anything else is refused
function prior_descriptors(d::Distribution)
    comps = hasproperty(d, :v) ? getproperty(d, :v) : hasproperty(d, :dists) ? getproperty(d, :dists) :
            error("prior must be Normal, Uniform, Exponential, LogNormal, Gamma, Beta, truncated(Normal), product_distribution([...]) of those, or MvNormal")
    [prior_descriptor(c) for c in comps]
end

# ANY other Distribution (SimulatedAnnealingABC.jl:151): rand(prior) (:174) and logpdf(prior, θ) (:314, :318) stay Julia calls,
# handed to the library as two host callbacks (sabc_set_host_prior, prior_joint = 2) -- next to a Julia `f_dist`, where the
# per-particle body is already cut at the host, or next to a device-coded one, which then runs as its own launch between the
# proposal and the accept kernel.
is_data_prior(d::Distribution) =
    d isa MvNormal || d isa Union{Normal,Uniform,Exponential,LogNormal,Gamma,Beta,Truncated{<:Normal}} ||
    ((hasproperty(d, :v) || hasproperty(d, :dists)) &&
     all(c -> c isa Union{Normal,Uniform,Exponential,LogNormal,Gamma,Beta,Truncated{<:Normal}}, hasproperty(d, :v) ? d.v : d.dists))

function host_prior_callbacks(prior::Distribution)
    d = length(prior)
    function sample_cb(ctx::Ptr{Cvoid}, m::Int64, ids::Ptr{Int64}, theta::Ptr{Float64})::Cint
        try
            Θ = unsafe_wrap(Array, theta, (Int(m), d))                # column-major m x d
            for i in 1:m
                Θ[i, :] .= rand(prior)
            end
            return Cint(0)
        catch
            return Cint(-1)
        end
    end
    function logpdf_cb(ctx::Ptr{Cvoid}, m::Int64, theta::Ptr{Float64}, lp::Ptr{Float64})::Cint
        try
            Θ = unsafe_wrap(Array, theta, (Int(m), d))
            L = unsafe_wrap(Array, lp, (Int(m),))
            Threads.@threads for i in 1:m
                L[i] = d == 1 ? logpdf(prior, Θ[i, 1]) : logpdf(prior, Θ[i, :])
            end
            return Cint(0)
        catch
            return Cint(-1)
        end
    end
    (@cfunction($sample_cb, Cint, (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Float64})),
     @cfunction($logpdf_cb, Cint, (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64})))
end

# ---- result types: same field names as SimulatedAnnealingABC.jl:28-60 ----
mutable struct SABCstate
    ϵ::Vector{Float64}
    algorithm::Symbol
    ϵ_history::Vector{Vector{Float64}}
    ρ_history::Vector{Vector{Float64}}
    u_history::Vector{Vector{Float64}}
    cdfs_dist_prior
    n_simulation::Int
    n_accept::Int
    n_resampling::Int
    n_population_updates::Int
end

struct SABCresult{T,S}                  # the reference's four fields, nothing else (:55-60)
    population::Vector{T}
    u::Array{S}
    ρ::Array{S}
    state::SABCstate
end

# The device-resident population behind a result: a side table keyed by the (mutable) state object, so that the handle
# lives exactly as long as the result and SABCresult keeps the reference's constructor.
const HANDLES = WeakKeyDict{SABCstate,Base.RefValue{Ptr{Cvoid}}}()
handle_of(res::SABCresult) = get(() -> error("this SABCresult has no device population (it was not created by sabc of SimulatedAnnealingABCHIP)"),
                                 HANDLES, res.state)

function check(h::Ptr{Cvoid}, rc::Integer)
    rc == 0 && return
    msg = unsafe_string(h == C_NULL ? ccall((:sabc_last_global_error, libsabc), Cstring, ()) :
                                      ccall((:sabc_last_error, libsabc), Cstring, (Ptr{Cvoid},), h))
    error(msg)                           # ErrorException, like the reference's error(...) sites
end

padtuple(v, n, T) = ntuple(i -> i <= length(v) ? T(v[i]) : zero(T), n)

"""
    comm_unique_id() -> Vector{UInt8}

The 128-byte RCCL id of a multi-GPU run: rank 0 calls this and hands the bytes to the other ranks (MPI.jl `bcast`, a
file, a socket ...); every rank then passes them as `comm_id` to `sabc`.
"""
function comm_unique_id()
    id = Vector{UInt8}(undef, 128)
    check(C_NULL, ccall((:sabc_comm_unique_id, libsabc), Cint, (Ptr{Cvoid},), id))
    id
end

function create_handle(f_dist::DeviceDistance, prior; n_particles, algorithm, v, δ, seed, device=0, rank=0, world=1,
                       comm_id=nothing, p2p=true)
    source_prior = prior isa SourcePrior
    source_prior && !(f_dist isa DeviceSource) && error("a SourcePrior is device code inside the HIP source of a DeviceSource")
    host_prior = !source_prior && !is_data_prior(prior)
    pd = (host_prior || source_prior) ? [(Int32(0), 0.0, 1.0, 0.0, 0.0) for _ in 1:length(prior)] : prior_descriptors(prior)
    joint, chol = host_prior ? (Int32(2), Float64[]) : source_prior ? (Int32(3), Float64[]) : prior_chol(prior)
    p = params(f_dist)
    cfg = Ref(CConfig(6, device, n_particles, length(pd), n_stats(f_dist), model_id(f_dist), length(p),
                      padtuple(p, MAX_MODEL_PARAMS, Float64),
                      padtuple(first.(pd), MAX_PARA, Int32),
                      padtuple(getindex.(pd, 2), MAX_PARA, Float64), padtuple(getindex.(pd, 3), MAX_PARA, Float64),
                      padtuple(getindex.(pd, 4), MAX_PARA, Float64), padtuple(getindex.(pd, 5), MAX_PARA, Float64),
                      joint, Int32(0), padtuple(chol, MAX_PARA2, Float64),
                      algorithm == :multi_eps ? 1 : 0, rank, world, 0, v, δ, seed))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(C_NULL, ccall((:sabc_create, libsabc), Cint, (Ref{CConfig}, Ref{Ptr{Cvoid}}), cfg, h))
    # A finalizer runs whenever the collector pleases, at a different time on every rank: it must neither free memory a peer
    # shard may be reading nor wait for peers.  sabc_destroy takes care of the first in any case (the shard LEAVES its
    # peer-to-peer group in order, include/sabc_hip.h "LEAVING"); a destroy wait of 0 makes it park what a peer has not
    # released instead of waiting for it.  `close(res)` is the orderly way out: it waits (bounded) and frees.
    finalizer(h) do r
        if r[] != C_NULL
            ccall((:sabc_comm_p2p_set_destroy_wait, libsabc), Cint, (Ptr{Cvoid}, Cdouble), r[], 0.0)
            ccall((:sabc_destroy, libsabc), Cvoid, (Ptr{Cvoid},), r[])
            delete!(HOST_CALLBACKS, r[])               # the library can no longer call back: release the closure
            r[] = C_NULL
        end
    end
    if world > 1                                       # one process per GPU: RCCL bound inside the library
        (comm_id isa Vector{UInt8} && length(comm_id) == 128) ||
            error("world > 1 needs `comm_id`: the 128 bytes of comm_unique_id() from rank 0")
        GC.@preserve comm_id check(h[], ccall((:sabc_comm_init_rccl, libsabc), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), h[], comm_id))
        check(h[], ccall((:sabc_comm_selftest, libsabc), Cint, (Ptr{Cvoid},), h[]))
        # on top of RCCL: the peer-to-peer transport (the shards of one node exchange through each other's HBM: one launch per
        # population update instead of reduce -> allreduce -> control).  ONE collective call: the descriptors travel over
        # the RCCL allgather just installed, every rank maps its peers and runs the self-test, and the ranks AGREE inside the
        # library after every step -- either all of them now run peer to peer (1) or all of them stay on RCCL (0): no rank
        # switches alone and leaves the others to wait out the bound of their first exchange.
        if p2p && world <= 8 && !(f_dist isa HostDistance)
            rc = ccall((:sabc_comm_p2p_setup, libsabc), Cint, (Ptr{Cvoid},), h[])
            rc < 0 && check(h[], rc)
        end
    end
    if f_dist isa DeviceSource
        check(h[], ccall((:sabc_register_device_simulator, libsabc), Cint, (Ptr{Cvoid}, Cstring), h[], f_dist.source))
    end
    if f_dist isa HostDistance
        cb = host_callback(f_dist)
        pcb = host_prior ? host_prior_callbacks(prior) : nothing
        HOST_CALLBACKS[h[]] = (cb, f_dist, pcb, prior) # keep the closures (and what they wrap) alive with the handle
        check(h[], ccall((:sabc_set_host_simulator, libsabc), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), h[], cb, C_NULL))
        if host_prior
            check(h[], ccall((:sabc_set_host_prior, libsabc), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                             h[], pcb[1], pcb[2], C_NULL))
        end
    elseif host_prior                                  # any Distribution next to a device-coded simulator
        pcb = host_prior_callbacks(prior)
        HOST_CALLBACKS[h[]] = (nothing, f_dist, pcb, prior)
        check(h[], ccall((:sabc_set_host_prior, libsabc), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                         h[], pcb[1], pcb[2], C_NULL))
    end
    h
end
const HOST_CALLBACKS = Dict{Ptr{Cvoid},Any}()

# the reference's own signature (SimulatedAnnealingABC.jl:451): any Function; the number of statistics is
# probed with one call on a prior draw, as at :163-165
function sabc(f_dist::Function, prior::Distribution, args...; kwargs...)
    f_dist isa DeviceDistance && return invoke(sabc, Tuple{DeviceDistance,Distribution}, f_dist, prior; kwargs...)
    own = (:n_particles, :n_simulation, :algorithm, :proposal, :resample, :v, :δ, :checkpoint_history,
           :show_progressbar, :show_checkpoint, :seed, :device, :rank, :world, :comm_id)
    mine = (; (k => v for (k, v) in kwargs if k in own)...)
    theirs = (; (k => v for (k, v) in kwargs if !(k in own))...)
    ρ = f_dist(rand(prior), args...; theirs...)
    hd = HostDistance(f_dist, length(ρ), length(prior), args, theirs)
    invoke(sabc, Tuple{DeviceDistance,Distribution}, hd, prior; mine...)
end

# copies device state into the Julia arrays in place (`.=` at SimulatedAnnealingABC.jl:395-397)
function refresh!(res::SABCresult, d::Int, s::Int)
    h = handle_of(res)[]
    n = length(res.population)
    θ = Matrix{Float64}(undef, n, d)               # column-major n x d == the library's [d][n]
    GC.@preserve θ check(h, ccall((:sabc_get_population, libsabc), Cint,
                                  (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), h, θ, res.u, res.ρ))
    if eltype(res.population) <: Real
        res.population .= vec(θ)
    else
        for i in 1:n
            res.population[i] = θ[i, :]
        end
    end
    st = res.state
    eps = Vector{Float64}(undef, MAX_STATS); len = Ref{Int32}(0)
    check(h, ccall((:sabc_get_epsilon, libsabc), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ref{Int32}), h, eps, len))
    st.ϵ = eps[1:len[]]
    c = Vector{Int64}(undef, 4)
    check(h, ccall((:sabc_get_counters, libsabc), Cint, (Ptr{Cvoid}, Ptr{Int64}), h, c))
    st.n_simulation, st.n_accept, st.n_resampling, st.n_population_updates = c
    m = ccall((:sabc_history_len, libsabc), Int64, (Ptr{Cvoid},), h)
    le = length(st.ϵ)
    eh = Matrix{Float64}(undef, le, m); uh = Matrix{Float64}(undef, s, m); rh = Matrix{Float64}(undef, s, m)
    check(h, ccall((:sabc_get_history, libsabc), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), h, eh, uh, rh))
    st.ϵ_history = [eh[:, k] for k in 1:m]; st.u_history = [uh[:, k] for k in 1:m]; st.ρ_history = [rh[:, k] for k in 1:m]
    res
end

# Check if a stream is logged (SimulatedAnnealingABC.jl:500)
is_logging(io) = isa(io, Base.TTY) == false || (get(ENV, "CI", nothing) == "true")

# Progress output needs the device loop to come up for air: `update_population!` is cut into several sabc_update calls.
# Returns the update counts after which a call ends: every multiple of `show_checkpoint` (:359), every step of the progress
# bar (a fiftieth of the run), and n_pop.  The cuts may fall anywhere: each call is told how many updates of the loop came
# before it (history_phase) and whether another follows (more_chunks_follow), so `ix % checkpoint_history` (:367) and the
# final push (:378-382) see the loop's own numbering -- `show_checkpoint` and `checkpoint_history` are independent moduli,
# as in the reference.  Same rule as progress_stops() in ../api.py.
function progress_stops(n_pop, show_checkpoint, show_progressbar)
    stops = Set{Int}([n_pop])
    if isfinite(show_checkpoint) && show_checkpoint >= 1
        union!(stops, Int(show_checkpoint):Int(show_checkpoint):(n_pop - 1))
    end
    if show_progressbar && n_pop > 0
        union!(stops, max(n_pop ÷ 50, 1):max(n_pop ÷ 50, 1):(n_pop - 1))
    end
    sort!(collect(stops))
end

# Where the next sabc_update call ends.  A call costs ~65 us beyond its updates, and ProgressMeter redraws every 0.1 s at most:
# progress-bar stops closer than 0.1 s of work (rate = population updates per second on the previous call, 0: unknown) are
# passed over; multiples of `show_checkpoint` and n_pop never are.  Same rule as next_stop() in ../api.py.
function next_stop(stops, n_pop, show_checkpoint, done, rate)
    target = done + (rate > 0 ? max(1, floor(Int, rate * 0.1)) : 1)
    chk = (isfinite(show_checkpoint) && show_checkpoint >= 1) ? Int(show_checkpoint) : 0
    for s in stops
        s <= done && continue
        (s >= target || s == n_pop || (chk > 0 && s % chk == 0)) && return s
    end
    n_pop
end

"""
    close(res::SABCresult)

Release the device population behind `res` now instead of whenever the collector finalizes it.  On a multi-GPU run this is
the orderly way out: the shard leaves its peer-to-peer group, waits (bounded) until its peers have unmapped its memory, and
frees it.  No barrier with the other ranks is needed -- a rank that is still inside `update_population!` when a peer closes
ends that call over RCCL (or with an error if RCCL is gone too), never with a memory fault.
"""
function Base.close(res::SABCresult)
    r = handle_of(res)
    if r[] != C_NULL
        ccall((:sabc_destroy, libsabc), Cvoid, (Ptr{Cvoid},), r[])
        delete!(HOST_CALLBACKS, r[])
        r[] = C_NULL
    end
    nothing
end

"""
    update_population!(population_state, f_dist, prior; n_simulation, v=1.0, δ=0.1, proposal, resample, checkpoint_history=1,
                       show_progressbar, show_checkpoint)

Same keywords and defaults as SimulatedAnnealingABC.jl:251-259.  Mutates and returns `population_state`.
"""
function update_population!(res::SABCresult, f_dist::DeviceDistance, prior::Distribution;
                            n_simulation, v=1.0, δ=0.1,
                            proposal::Proposal=DifferentialEvolution(n_para=length(prior)),
                            resample=nothing, checkpoint_history=1,
                            show_progressbar::Bool=!is_logging(stderr),
                            show_checkpoint=is_logging(stderr) ? 100 : Inf)
    v <= 0 && error("Annealing speed `v` must be positive.")                       # :261
    δ <= 0 && error("Resamping intensity `δ` must be positive.")                   # :262
    h = handle_of(res)[]
    d, s = length(prior), size(res.u, 2)
    n_global = ccall((:sabc_n_global, libsabc), Int64, (Ptr{Cvoid},), h)           # == length(population) on one GPU
    resample = something(resample, 2 * n_global)                                   # :255
    θ = eltype(res.population) <: Real ? reshape(copy(res.population), :, 1) : permutedims(reduce(hcat, res.population))
    GC.@preserve θ check(h, ccall((:sabc_set_population, libsabc), Cint,
                                  (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), h, θ, res.u, res.ρ))
    kind, p0, p1 = descriptor(proposal)
    n_pop = n_simulation ÷ n_global                                                # :275
    n_pop > 0 && checkpoint_history == 0 && throw(DivideError())                   # `ix % checkpoint_history`, :367
    pmeter = Progress(n_pop; desc="$n_pop population updates:", output=stderr, enabled=show_progressbar)   # :290-291
    t_start = Dates.now()
    done = 0
    stops = progress_stops(n_pop, show_checkpoint, show_progressbar)
    rate, first = 0.0, true
    while first || done < n_pop                                                    # (n_pop = 0: one call)
        first = false
        stop = next_stop(stops, n_pop, show_checkpoint, done, rate)
        todo = stop - done
        budget = n_pop > 0 ? todo * n_global : n_simulation                        # a top-up below one update is a no-op (:275)
        args = Ref(CUpdateArgs(budget, v, δ, resample, checkpoint_history, kind, stop < n_pop ? 1 : 0, p0, p1, done))
        t_call = time()
        check(h, ccall((:sabc_update, libsabc), Cint, (Ptr{Cvoid}, Ref{CUpdateArgs}), h, args))
        rate = todo / max(time() - t_call, 1e-9)
        done = stop
        if show_progressbar || (isfinite(show_checkpoint) && show_checkpoint >= 1)
            eps = Vector{Float64}(undef, MAX_STATS); len = Ref{Int32}(0)
            check(h, ccall((:sabc_get_epsilon, libsabc), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ref{Int32}), h, eps, len))
            ϵ = round.(eps[1:len[]], sigdigits=4)
            show_progressbar && next!(pmeter; step=todo, showvalues=[("ϵ", ϵ)])    # :292,374
            if done > 0 && isfinite(show_checkpoint) && show_checkpoint >= 1 && done % Int(show_checkpoint) == 0   # :359-364
                eta = ((Dates.now() - t_start) ÷ done) * (n_pop - done)
                etastr = eta > Dates.Second(1) ? Dates.canonicalize(round(eta, Dates.Second)) : "< 1 Second"
                @info "Update $done of $n_pop. ϵ: $ϵ, ETA: $(etastr)"; flush(stderr)
            end
        end
    end
    show_progressbar && finish!(pmeter)
    if proposal isa RandomWalk
        Σ = Matrix{Float64}(undef, d, d)
        check(h, ccall((:sabc_get_proposal_sigma, libsabc), Cint, (Ptr{Cvoid}, Ptr{Float64}), h, Σ))
        d == 1 ? (proposal.Σ = Σ[1, 1]) : (proposal.Σ .= Σ)
    end
    refresh!(res, d, s)
    @info "All particles have been updated $(n_pop) times."; flush(stderr)          # :399
    res
end

# update_population!(res, f_dist, prior, args...; kwargs...) with the plain function the result was created
# with (SimulatedAnnealingABC.jl:251): look the wrapped model up by handle; args/kwargs go to f_dist (:315)
function update_population!(res::SABCresult, f_dist::Function, prior::Distribution, args...; kwargs...)
    f_dist isa DeviceDistance && return invoke(update_population!, Tuple{SABCresult,DeviceDistance,Distribution}, res, f_dist, prior; kwargs...)
    haskey(HOST_CALLBACKS, handle_of(res)[]) || error("this SABCresult was not created with a host `f_dist`")
    hd = HOST_CALLBACKS[handle_of(res)[]][2]
    (hd isa HostDistance && hd.f === f_dist) || error("`f_dist` differs from the one this SABCresult was initialised with")
    own = (:n_simulation, :v, :δ, :proposal, :resample, :checkpoint_history, :show_progressbar, :show_checkpoint)
    mine = (; (k => v for (k, v) in kwargs if k in own)...)
    invoke(update_population!, Tuple{SABCresult,DeviceDistance,Distribution}, res, hd, prior; mine...)
end

"""
    sabc(f_dist::DeviceDistance, prior; n_particles=100, n_simulation=10_000, algorithm=:single_eps, ...)

Same keywords as SimulatedAnnealingABC.jl:451-460, plus `seed` (Philox key), `device` and -- one process per GPU --
`rank`, `world`, `comm_id` (see `comm_unique_id`).  On a sharded run `n_particles` is the GLOBAL count, the result holds
this rank's shard and every rank must pass the same `seed`.
"""
function sabc(f_dist::DeviceDistance, prior::Distribution;
              n_particles=100, n_simulation=10_000, algorithm=:single_eps,
              proposal::Proposal=DifferentialEvolution(n_para=length(prior)),
              resample=2 * n_particles, v=1.0, δ=0.1, checkpoint_history=1,
              show_progressbar::Bool=!is_logging(stderr), show_checkpoint=is_logging(stderr) ? 100 : Inf,
              seed=nothing, device=0, rank=0, world=1, comm_id=nothing)
    (algorithm == :multi_eps || algorithm == :single_eps) ||
        error("Argument `algorithm` must be :multi_eps or :single_eps, not `$algorithm`!")   # :462-464
    n_simulation < n_particles &&
        error("`n_simulation = $n_simulation` is too small for $n_particles particles.")     # :155-156
    world > 1 && isnothing(seed) && error("world > 1 needs an explicit `seed`: every shard must use the same Philox key")
    seed = something(seed, rand(UInt64) >> 1)
    @info "Initialization for '$(algorithm)'"                                                # :158
    h = create_handle(f_dist, prior; n_particles, algorithm, v, δ, seed, device, rank, world, comm_id)
    check(h[], ccall((:sabc_initialize, libsabc), Cint, (Ptr{Cvoid}, Int64), h[], n_simulation))
    d, s = length(prior), n_stats(f_dist)
    n_local = ccall((:sabc_n_local, libsabc), Int64, (Ptr{Cvoid},), h[])
    T = d == 1 ? Float64 : Vector{Float64}
    pop = d == 1 ? zeros(n_local) : [zeros(d) for _ in 1:n_local]
    cdf = ρ -> begin
        out = Vector{Float64}(undef, s)
        r = collect(Float64, ρ)
        check(h[], ccall((:sabc_cdf_apply, libsabc), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Float64}), h[], r, 1, out))
        out
    end
    st = SABCstate(Float64[], algorithm, [], [], [], cdf, 0, 0, 0, 0)
    HANDLES[st] = h
    res = SABCresult{T,Float64}(pop, zeros(n_local, s), zeros(n_local, s), st)
    refresh!(res, d, s)
    n_sim_remaining = n_simulation - st.n_simulation                                         # :478
    n_sim_remaining < n_particles && @warn "`n_simulation` too small to update all particles!"
    update_population!(res, f_dist, prior; n_simulation=n_sim_remaining, resample, proposal, v, δ, checkpoint_history,
                       show_progressbar, show_checkpoint)
end

function show(io::IO, s::SABCresult)     # SimulatedAnnealingABC.jl:65-82
    n = length(s.population)
    println(io, "Approximate posterior sample with $n particles:")
    println(io, "  - algorithm: :$(s.state.algorithm)")
    println(io, "  - simulations used: $(s.state.n_simulation)")
    println(io, "  - number of population updates: $(s.state.n_population_updates)")
    println(io, "  - ϵ: $(round.(s.state.ϵ, sigdigits=4))")
    println(io, "  - number of population resamplings: $(s.state.n_resampling)")
    println(io, "  - acceptance rate: $(round(s.state.n_accept / (s.state.n_simulation - n), sigdigits=4))")
end

end # module
