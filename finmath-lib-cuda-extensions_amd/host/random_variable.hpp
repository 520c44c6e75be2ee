// random_variable.hpp — C++ host-side mirror of the reference's plug-in interfaces, above the C-ABI.
//
//   reference (Java)                                              here (namespace fmhost)
//   ------------------------------------------------------------------------------------------------------
//   net.finmath.stochastic.RandomVariable (finmath-lib, external) RandomVariable   (abstract interface)
//   net.finmath.montecarlo.RandomVariableFactory                  RandomVariableFactory
//   net.finmath.montecarlo.BrownianMotion                         BrownianMotion
//   RandomVariableCuda        (RandomVariableCuda.java)           RandomVariableHip
//   RandomVariableCudaFactory (RandomVariableCudaFactory.java)    RandomVariableHipFactory
//   BrownianMotionCudaWithRandomVariableCuda                      BrownianMotionHip
//
// Same method names, argument meaning, dispatch order (type priority → newTime → deterministic fast paths) and
// error behaviour as RandomVariableCuda (`:line` cites RandomVariableCuda.java; the deviations that follow the
// reference's own CPU twin are the ones listed in ../random_variable.py).  Random variables are immutable and
// shared: `RV` = std::shared_ptr<const RandomVariable>.  Models (lmm.hpp) are written against the interfaces
// only, so the same model code runs on any implementation — exactly how finmath-lib injects a factory
// (LIBORMarketModelCalibrationATMTest.java:351-358).
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <functional>
#include <limits>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fmhip.h"
#include "mersenne.hpp"

namespace fmhost {

class RandomVariable;
using RV = std::shared_ptr<const RandomVariable>;

// ------------------------------------------------------------------ interfaces

class RandomVariable : public std::enable_shared_from_this<RandomVariable> {
public:
    virtual ~RandomVariable() = default;
    // accessors
    virtual double getFiltrationTime() const = 0;
    virtual int    getTypePriority() const = 0;
    virtual bool   isDeterministic() const = 0;
    virtual int64_t size() const = 0;
    virtual double doubleValue() const = 0;                      // throws if stochastic (:1125-1131)
    virtual std::vector<double> getRealizations() const = 0;     // :1116-1123
    // reductions
    virtual double getAverage() const = 0;
    virtual double getVariance() const = 0;
    virtual double getMin() const = 0;
    virtual double getMax() const = 0;
    // paths behind an expectation: size() unless the implementation shards its paths over processes (RandomVariableHip with an
    // expectation communicator: size() of this process's shard times the number of ranks)
    virtual int64_t sampleSize() const { return size(); }
    double getSampleVariance() const { const int64_t n = sampleSize(); return (isDeterministic() || size() == 1) ? 0.0 : getVariance() * n / (n - 1); }
    double getStandardDeviation() const { return isDeterministic() ? 0.0 : std::sqrt(getVariance()); }
    double getStandardError() const { return isDeterministic() ? 0.0 : getStandardDeviation() / std::sqrt((double)sampleSize()); }
    virtual double getAverage(const RV& probabilities) const { return mult(probabilities)->getAverage(); }   // :886-888
    // scalar operand / unary
    virtual RV cap(double v) const = 0;
    virtual RV floor(double v) const = 0;
    virtual RV add(double v) const = 0;
    virtual RV sub(double v) const = 0;
    virtual RV bus(double v) const = 0;
    virtual RV mult(double v) const = 0;
    virtual RV div(double v) const = 0;
    virtual RV vid(double v) const = 0;
    virtual RV pow(double e) const = 0;
    virtual RV squared() const = 0;
    virtual RV sqrt() const = 0;
    virtual RV exp() const = 0;
    virtual RV log() const = 0;
    virtual RV invert() const = 0;
    virtual RV abs() const = 0;
    virtual RV isNaN() const = 0;
    virtual RV sin() const = 0;                                     // GPU class throws (:1373-1384); semantics from the twin (:927-954)
    virtual RV cos() const = 0;
    // host-side cold paths: the reference sorts / maps on the host as well (:970-1091; twin :473-602, :667-748)
    virtual RV apply(const std::function<double(double)>& f) const = 0;
    double getQuantile(double quantile) const;                      // index convention of the GPU class (:983: 1 - quantile)
    double getQuantileExpectation(double quantileStart, double quantileEnd) const;
    std::vector<double> getHistogram(const std::vector<double>& intervalPoints) const;                     // :1026-1068
    std::vector<std::vector<double>> getHistogram(int numberOfPoints, double standardDeviations) const;    // :1070-1091
    // vector operand
    virtual RV cap(const RV& rv) const = 0;
    virtual RV floor(const RV& rv) const = 0;
    virtual RV add(const RV& rv) const = 0;
    virtual RV sub(const RV& rv) const = 0;
    virtual RV bus(const RV& rv) const = 0;
    virtual RV mult(const RV& rv) const = 0;
    virtual RV div(const RV& rv) const = 0;
    virtual RV vid(const RV& rv) const = 0;
    virtual RV accrue(const RV& rate, double periodLength) const = 0;
    virtual RV discount(const RV& rate, double periodLength) const = 0;
    virtual RV choose(const RV& valueIfTriggerNonNegative, const RV& valueIfTriggerNegative) const = 0;
    virtual RV addProduct(const RV& factor1, double factor2) const = 0;
    virtual RV addProduct(const RV& factor1, const RV& factor2) const = 0;
    virtual RV addRatio(const RV& numerator, const RV& denominator) const { return add(numerator->div(denominator)); }   // :1686-1689
    virtual RV subRatio(const RV& numerator, const RV& denominator) const { return sub(numerator->div(denominator)); }   // :1692-1695
    RV average() const;                                           // :1280
    RV self() const { return shared_from_this(); }
};

class RandomVariableFactory {
public:
    virtual ~RandomVariableFactory() = default;
    virtual RV createRandomVariable(double value) const = 0;                                // time = -infinity
    virtual RV createRandomVariable(double time, double value) const = 0;                   // RandomVariableCudaFactory.java:27
    virtual RV createRandomVariable(double time, const std::vector<double>& values) const = 0;   // :32
};

class TimeDiscretization {      // stand-in for net.finmath.time.TimeDiscretizationFromArray (finmath-lib, not vendored)
public:
    TimeDiscretization() = default;
    TimeDiscretization(double initial, int numberOfTimeSteps, double deltaT) {
        for (int i = 0; i <= numberOfTimeSteps; ++i) times_.push_back(initial + i * deltaT);
    }
    explicit TimeDiscretization(std::vector<double> times) : times_(std::move(times)) {}
    int getNumberOfTimeSteps() const { return (int)times_.size() - 1; }
    int getNumberOfTimes() const { return (int)times_.size(); }
    double getTime(int i) const { return times_.at((size_t)i); }
    double getTimeStep(int i) const { return times_.at((size_t)i + 1) - times_.at((size_t)i); }
    // getTimeIndex: index of `time`, or -(insertion point)-1 (java.util.Arrays.binarySearch contract)
    int getTimeIndex(double time) const {
        int lo = 0, hi = (int)times_.size() - 1;
        while (lo <= hi) { const int mid = (lo + hi) / 2; if (std::fabs(times_[mid] - time) < 1e-12) return mid; if (times_[mid] < time) lo = mid + 1; else hi = mid - 1; }
        return -(lo) - 1;
    }
    int getTimeIndexNearestLessOrEqual(double time) const { int i = getTimeIndex(time); if (i < 0) i = -i - 2; return i; }
    const std::vector<double>& asVector() const { return times_; }
private:
    std::vector<double> times_;
};

class BrownianMotion {
public:
    virtual ~BrownianMotion() = default;
    virtual RV getBrownianIncrement(int timeIndex, int factor) const = 0;
    virtual const TimeDiscretization& getTimeDiscretization() const = 0;
    virtual int getNumberOfFactors() const = 0;
    virtual int64_t getNumberOfPaths() const = 0;
    virtual RV getRandomVariableForConstant(double value) const = 0;
    RV getIncrement(int timeIndex, int factor) const { return getBrownianIncrement(timeIndex, factor); }    // IndependentIncrements (:205-208)
};

// ------------------------------------------------------------------ helpers shared by implementations

inline double jmin(double a, double b) { if (a != a) return a; if (a == 0.0 && b == 0.0 && std::signbit(b)) return b; return (a <= b) ? a : b; }
inline double jmax(double a, double b) { if (a != a) return a; if (a == 0.0 && b == 0.0 && std::signbit(a)) return b; return (a >= b) ? a : b; }
inline double jpow(double x, double y) { if (y == 0.0) return 1.0; if (y != y) return y; if (std::isinf(y) && std::fabs(x) == 1.0) return std::nan(""); return std::pow(x, y); }

struct UnsupportedOperation : std::logic_error { using std::logic_error::logic_error; };

// ------------------------------------------------------------------ device vector (RAII over one fmhip_vec handle)

struct FmhipError : std::runtime_error {
    int code;
    FmhipError(int c, const std::string& m) : std::runtime_error("fmhip error " + std::to_string(c) + ": " + m), code(c) {}
};
inline void check(int status) { if (status != FMHIP_OK) throw FmhipError(status, fmhip_last_error()); }

// ------------------------------------------------------------------ a garbage collector's idea of a handle's lifetime
//
// The Java binding releases a handle when the collector says so: java/net/finmath/hip/DeviceVector.java registers a Cleaner action
// per handle, the reference recycles device pointers through WeakReference / ReferenceQueue (RandomVariableCuda.java:96-106,
// 293-305, 384-385).  The temporary of `x.add(y).mult(z)` is unreachable after the statement, but its release arrives at the NEXT
// collection — tens to hundreds of milliseconds and thousands of recorded methods later —, in a burst, on the collector's thread.
// This C++ mirror destroys temporaries at the end of the expression (RAII); ReleaseLag gives it the JVM's behaviour for measurements
// and tests (lmm_hip --release-lag, tests/cpp/test_release_lag.cpp): every release the mirror would issue is queued instead and
//   collectEveryMs > 0   a collector thread wakes every so many milliseconds and releases everything that was queued when it woke
//                        (a periodic young collection: a dead object waits between 0 and one period);
//   collectAtBytes > 0   … or releases everything once the dead wrappers amount to that many bytes of JVM heap (BYTES_PER_HANDLE each:
//                        DeviceVector + its Cleaner registration + the RandomVariable around it) — "never, until the young
//                        generation is full";
//   a device allocation that fails (FMHIP_ERR_OUT_OF_MEMORY) runs a collection on the calling thread and is tried again once, as the
//   reference does when device memory runs short (System.gc(), RandomVariableCuda.java:311-335).
class ReleaseLag {
public:
    static constexpr size_t BYTES_PER_HANDLE = 128;
    static ReleaseLag& instance() { static ReleaseLag r; return r; }
    struct Stats { long long queued = 0, released = 0, collections = 0, forcedCollections = 0, peakQueue = 0; };
    bool on() const { return on_.load(std::memory_order_relaxed); }
    void start(double collectEveryMs, size_t collectAtBytes) {
        stop();
        if (!(collectEveryMs > 0.0) && collectAtBytes == 0) return;
        everyMs_ = collectEveryMs; atBytes_ = collectAtBytes; quit_.store(false);
        on_.store(true);
        collector_ = std::thread([this] { run(); });
    }
    void stop() {                                        // the JVM exits: whatever is queued is released
        if (!on_.load()) return;
        quit_.store(true);
        if (collector_.joinable()) collector_.join();
        on_.store(false);
        collect(false);
    }
    // A wrapper object has become unreachable.  (A JVM pays nothing for this moment; the model pays a push into a buffer of the thread's
    // own, handed to the shared list 256 at a time — the collector does not see the last few of a thread that has gone quiet, as it would
    // not see objects that still sit in a register.)
    void died(fmhip_vec h) {
        std::vector<fmhip_vec>& mine = local();
        mine.push_back(h);
        if (mine.size() >= 256) publish(mine);
    }
    void collect(bool forced) {                          // one collection, on the calling thread
        publish(local());
        std::vector<fmhip_vec> batch;
        { std::lock_guard<std::mutex> lock(mu_); batch.swap(dead_); full_.store(false, std::memory_order_release); ++stats_.collections; if (forced) ++stats_.forcedCollections; stats_.released += (long long)batch.size(); }
        for (fmhip_vec h : batch) fmhip_vec_release(h);
    }
    Stats stats() { std::lock_guard<std::mutex> lock(mu_); return stats_; }
    ~ReleaseLag() { quit_.store(true); if (collector_.joinable()) collector_.join(); }
private:
    static std::vector<fmhip_vec>& local() { static thread_local std::vector<fmhip_vec> v; return v; }
    void publish(std::vector<fmhip_vec>& mine) {
        if (mine.empty()) return;
        std::lock_guard<std::mutex> lock(mu_);
        dead_.insert(dead_.end(), mine.begin(), mine.end());
        stats_.queued += (long long)mine.size();
        mine.clear();
        stats_.peakQueue = std::max<long long>(stats_.peakQueue, (long long)dead_.size());
        if (atBytes_ > 0 && dead_.size() * BYTES_PER_HANDLE >= atBytes_) full_.store(true, std::memory_order_release);
    }
    void run() {                                         // (sleeps in slices and looks at two flags: no condition variable — the sanitizer builds' libtsan does not know pthread_cond_clockwait)
        using clk = std::chrono::steady_clock;
        auto next = clk::now() + std::chrono::duration_cast<clk::duration>(std::chrono::duration<double, std::milli>(everyMs_ > 0.0 ? everyMs_ : 1e12));
        while (!quit_.load(std::memory_order_acquire)) {
            std::this_thread::sleep_for(std::chrono::microseconds(everyMs_ > 0.0 && everyMs_ < 2.0 ? (long long)(everyMs_ * 500) : 1000));
            const bool due = everyMs_ > 0.0 && clk::now() >= next;
            if (!due && !full_.load(std::memory_order_acquire)) continue;
            collect(false);
            if (due) next = clk::now() + std::chrono::duration_cast<clk::duration>(std::chrono::duration<double, std::milli>(everyMs_));
        }
    }
    std::atomic<bool> on_{ false }, quit_{ false }, full_{ false };
    std::mutex mu_;
    std::vector<fmhip_vec> dead_;
    std::thread collector_;
    double everyMs_ = 0.0; size_t atBytes_ = 0;
    Stats stats_;
};

// Move-only owner of one handle, held BY VALUE inside a RandomVariableHip (no heap object of its own: a Monte-Carlo
// driver creates tens of thousands of random variables per objective evaluation and is bound by host time).  Sharing a
// vector between two random variables goes through the engine's own reference count (fmhip_vec_retain).
class DeviceVector {
public:
    DeviceVector() = default;
    explicit DeviceVector(fmhip_vec h) : h_(h) {}
    ~DeviceVector() { drop(); }
    DeviceVector(DeviceVector&& o) noexcept : h_(o.h_) { o.h_ = 0; }
    DeviceVector& operator=(DeviceVector&& o) noexcept { if (this != &o) { drop(); h_ = o.h_; o.h_ = 0; } return *this; }
    DeviceVector(const DeviceVector&) = delete;
    DeviceVector& operator=(const DeviceVector&) = delete;
    fmhip_vec handle() const { return h_; }
    bool valid() const { return h_ != 0; }
    DeviceVector share() const { if (!h_) return DeviceVector(); check(fmhip_vec_retain(h_)); return DeviceVector(h_); }
    static DeviceVector fromHost(const std::vector<double>& v) { fmhip_vec h = 0; guarded([&] { return fmhip_vec_create_from_double(v.data(), (int64_t)v.size(), &h); }); return DeviceVector(h); }
    static DeviceVector filled(int64_t n, double value) { fmhip_vec h = 0; guarded([&] { return fmhip_vec_create_filled(n, value, &h); }); return DeviceVector(h); }
    // the five launch helpers of the reference (callFunctionv1s0 … v3s0, :483-537), on raw handles
    static DeviceVector v1s0(int op, fmhip_vec a) { fmhip_vec o = 0; guarded([&] { return fmhip_call_v1s0(op, a, &o); }); return DeviceVector(o); }
    static DeviceVector v1s1(int op, fmhip_vec a, double s) { fmhip_vec o = 0; guarded([&] { return fmhip_call_v1s1(op, a, s, &o); }); return DeviceVector(o); }
    static DeviceVector v2s0(int op, fmhip_vec a, fmhip_vec b) { fmhip_vec o = 0; guarded([&] { return fmhip_call_v2s0(op, a, b, &o); }); return DeviceVector(o); }
    static DeviceVector v2s1(int op, fmhip_vec a, fmhip_vec b, double s) { fmhip_vec o = 0; guarded([&] { return fmhip_call_v2s1(op, a, b, s, &o); }); return DeviceVector(o); }
    static DeviceVector v3s0(int op, fmhip_vec a, fmhip_vec b, fmhip_vec c) { fmhip_vec o = 0; guarded([&] { return fmhip_call_v3s0(op, a, b, c, &o); }); return DeviceVector(o); }
    fmhip_moments moments(double shift = 0.0) const { fmhip_moments m; guarded([&] { return fmhip_reduce_moments(h_, shift, &m); }); return m; }
    // A call that may allocate device memory: out of memory under a lagging collector → collect, try once more (ReleaseLag).
    template <class Call> static void guarded(Call&& call) {
        int status = call();
        if (status == FMHIP_ERR_OUT_OF_MEMORY && ReleaseLag::instance().on()) { ReleaseLag::instance().collect(true); status = call(); }
        check(status);
    }
private:
    void drop() { if (!h_) return; if (ReleaseLag::instance().on()) ReleaseLag::instance().died(h_); else fmhip_vec_release(h_); h_ = 0; }
    fmhip_vec h_ = 0;
};

// ------------------------------------------------------------------ RandomVariableHip

class RandomVariableHip final : public RandomVariable {
public:
    static constexpr int typePriorityDefault = 20;                // :568
    // constant (:683-689)
    RandomVariableHip(double time, double value, int typePriority = typePriorityDefault)
        : time_(time), value_(value), n_(1), priority_(typePriority) {}
    // device vector (RandomVariableCuda.of, :618-646)
    RandomVariableHip(double time, DeviceVector&& realizations, int64_t n, int typePriority = typePriorityDefault)
        : time_(time), value_(std::numeric_limits<double>::quiet_NaN()), vec_(std::move(realizations)), n_(n), priority_(typePriority) {}
    // host values, narrowed to fp32 and uploaded (:696-723)
    RandomVariableHip(double time, const std::vector<double>& values)
        : time_(time), value_(std::numeric_limits<double>::quiet_NaN()), vec_(DeviceVector::fromHost(values)), n_((int64_t)values.size()), priority_(typePriorityDefault) {}

    static RV of(double time, double value) { return std::make_shared<RandomVariableHip>(time, value); }
    static RV restamp(const RV& rv, double time) {                   // same value, another filtration time
        const auto* h = dynamic_cast<const RandomVariableHip*>(rv.get());
        if (!h || h->time_ == time) return rv;
        return h->isDeterministic() ? of(time, h->value_) : of(time, h->vec_.share(), h->n_);
    }
    static RV of(double time, DeviceVector&& v, int64_t n) { return std::make_shared<RandomVariableHip>(time, std::move(v), n); }

    double getFiltrationTime() const override { return time_; }
    int getTypePriority() const override { return priority_; }
    bool isDeterministic() const override { return !vec_.valid(); }
    int64_t size() const override { return isDeterministic() ? 1 : n_; }
    double doubleValue() const override {
        if (isDeterministic()) return value_;
        throw UnsupportedOperation("The random variable is non-deterministic");
    }
    std::vector<double> getRealizations() const override {
        if (isDeterministic()) return { value_ };
        std::vector<double> out((size_t)n_);
        DeviceVector::guarded([&] { return fmhip_vec_read_double(vec_.handle(), out.data(), n_); });
        return out;
    }
    const DeviceVector& deviceVector() const { return vec_; }

    // ---- reductions on the device (replaces :830-901).  With an expectation communicator (fmhip_set_expectation_comm) the moments
    // are those of the GLOBAL vector (this process holds one shard of its paths) and the sample size is that of all ranks.
    int64_t sampleSize() const override { int world = 1; check(fmhip_expectation_world(&world, nullptr)); return size() * world; }
    double getAverage() const override {
        if (isDeterministic()) return value_;
        if (n_ == 0) return std::nan("");
        return vec_.moments().sum / (double)sampleSize();
    }
    double getVariance() const override {                          // twin two-pass Σ(x-mean)²/n (twin:360-382)
        if (isDeterministic() || n_ == 1) return 0.0;
        if (n_ == 0) return std::nan("");
        const double mean = getAverage();
        return vec_.moments(mean).sumsq / (double)sampleSize();
    }
    double getMin() const override { return isDeterministic() ? value_ : vec_.moments().min; }
    double getMax() const override { return isDeterministic() ? value_ : vec_.moments().max; }

    // ---- scalar operand / unary (:1172-1352)
    RV cap(double v) const override   { return scalar(FMHIP_OP_CAP_S, v, jmin(value_, v)); }
    RV floor(double v) const override { return scalar(FMHIP_OP_FLOOR_S, v, jmax(value_, v)); }
    RV add(double v) const override   { return scalar(FMHIP_OP_ADD_S, v, value_ + v); }
    RV sub(double v) const override   { return scalar(FMHIP_OP_SUB_S, v, value_ - v); }
    RV bus(double v) const override   { return scalar(FMHIP_OP_BUS_S, v, -value_ + v); }
    RV mult(double v) const override  { return scalar(FMHIP_OP_MULT_S, v, value_ * v); }
    RV div(double v) const override   { return scalar(FMHIP_OP_DIV_S, v, value_ / v); }
    RV vid(double v) const override   { return scalar(FMHIP_OP_VID_S, v, v / value_); }
    RV pow(double e) const override   { return scalar(FMHIP_OP_POW_S, e, jpow(value_, e)); }
    RV squared() const override { return unary(FMHIP_OP_SQUARED, value_ * value_); }
    RV sqrt() const override    { return unary(FMHIP_OP_SQRT, std::sqrt(value_)); }
    RV exp() const override     { return unary(FMHIP_OP_EXP, std::exp(value_)); }
    RV log() const override     { return unary(FMHIP_OP_LOG, std::log(value_)); }
    RV invert() const override  { return unary(FMHIP_OP_INVERT, 1.0 / value_); }
    RV abs() const override     { return unary(FMHIP_OP_ABS, std::fabs(value_)); }
    RV isNaN() const override   { return unary(FMHIP_OP_ISNAN, value_ != value_ ? 1.0 : 0.0); }
    RV sin() const override     { return unary(FMHIP_OP_SIN, std::sin(value_)); }
    RV cos() const override     { return unary(FMHIP_OP_COS, std::cos(value_)); }
    RV apply(const std::function<double(double)>& f) const override {       // twin :667-676: map on the host, narrow, new variable
        if (isDeterministic()) return of(time_, f(value_));
        std::vector<double> v = getRealizations();
        for (double& x : v) x = f(x);
        return of(time_, DeviceVector::fromHost(v), n_);
    }

    // ---- vector operand (:1391-1580)
    RV add(const RV& rv) const override {
        if (rv->getTypePriority() > priority_) return rv->add(self());
        const double t = std::max(time_, rv->getFiltrationTime());
        if (isDeterministic() && rv->isDeterministic()) return of(t, value_ + rv->doubleValue());
        if (isDeterministic()) return of(t, DeviceVector::v1s1(FMHIP_OP_ADD_S, vecOf(rv).h, value_), rv->size());
        if (rv->isDeterministic()) return of(t, DeviceVector::v1s1(FMHIP_OP_ADD_S, vec_.handle(), rv->doubleValue()), n_);
        return of(t, DeviceVector::v2s0(FMHIP_OP_ADD, vec_.handle(), vecOf(rv).h), n_);
    }
    RV sub(const RV& rv) const override {
        if (rv->getTypePriority() > priority_) return rv->bus(self());
        const double t = std::max(time_, rv->getFiltrationTime());
        if (isDeterministic() && rv->isDeterministic()) return of(t, value_ - rv->doubleValue());
        if (isDeterministic()) return of(t, DeviceVector::v1s1(FMHIP_OP_BUS_S, vecOf(rv).h, value_), rv->size());
        if (rv->isDeterministic()) return of(t, DeviceVector::v1s1(FMHIP_OP_SUB_S, vec_.handle(), rv->doubleValue()), n_);
        return of(t, DeviceVector::v2s0(FMHIP_OP_SUB, vec_.handle(), vecOf(rv).h), n_);
    }
    RV bus(const RV& rv) const override {
        if (rv->getTypePriority() > priority_) return rv->sub(self());
        const double t = std::max(time_, rv->getFiltrationTime());
        if (isDeterministic() && rv->isDeterministic()) return of(t, -value_ + rv->doubleValue());
        if (isDeterministic()) return of(t, DeviceVector::v1s1(FMHIP_OP_SUB_S, vecOf(rv).h, value_), rv->size());
        if (rv->isDeterministic()) return of(t, DeviceVector::v1s1(FMHIP_OP_BUS_S, vec_.handle(), rv->doubleValue()), n_);
        return of(t, DeviceVector::v2s0(FMHIP_OP_SUB, vecOf(rv).h, vec_.handle()), n_);                   // flipped arguments, :1458
    }
    RV mult(const RV& rv) const override {
        if (rv->getTypePriority() > priority_) return rv->mult(self());
        const double t = std::max(time_, rv->getFiltrationTime());
        if (isDeterministic() && rv->isDeterministic()) return of(t, value_ * rv->doubleValue());
        if (rv->isDeterministic()) return mult(rv->doubleValue());
        if (isDeterministic()) return of(t, DeviceVector::v1s1(FMHIP_OP_MULT_S, vecOf(rv).h, value_), rv->size());
        return of(t, DeviceVector::v2s0(FMHIP_OP_MULT, vec_.handle(), vecOf(rv).h), n_);
    }
    RV div(const RV& rv) const override {
        if (rv->getTypePriority() > priority_) return rv->vid(self());
        const double t = std::max(time_, rv->getFiltrationTime());
        if (isDeterministic() && rv->isDeterministic()) return of(t, value_ / rv->doubleValue());
        if (isDeterministic()) return of(t, DeviceVector::v1s1(FMHIP_OP_VID_S, vecOf(rv).h, value_), rv->size());
        if (rv->isDeterministic()) return div(rv->doubleValue());
        return of(t, DeviceVector::v2s0(FMHIP_OP_DIV, vec_.handle(), vecOf(rv).h), n_);
    }
    RV vid(const RV& rv) const override {
        if (rv->getTypePriority() > priority_) return rv->div(self());             // twin:1116-1119
        const double t = std::max(time_, rv->getFiltrationTime());
        if (isDeterministic() && rv->isDeterministic()) return of(t, rv->doubleValue() / value_);
        if (isDeterministic()) return of(t, DeviceVector::v1s1(FMHIP_OP_DIV_S, vecOf(rv).h, value_), rv->size());
        if (rv->isDeterministic()) return restamp(vid(rv->doubleValue()), t);      // value as :1528, time as the twin (twin:1135-1140)
        return of(t, DeviceVector::v2s0(FMHIP_OP_DIV, vecOf(rv).h, vec_.handle()), n_);                   // flipped arguments, :1531
    }
    RV cap(const RV& rv) const override {
        if (rv->getTypePriority() > priority_) return rv->cap(self());
        const double t = std::max(time_, rv->getFiltrationTime());
        if (isDeterministic() && rv->isDeterministic()) return of(t, jmin(value_, rv->doubleValue()));
        if (isDeterministic()) return of(t, DeviceVector::v1s1(FMHIP_OP_CAP_S, vecOf(rv).h, value_), rv->size());
        if (rv->isDeterministic()) return of(t, DeviceVector::v1s1(FMHIP_OP_CAP_S, vec_.handle(), rv->doubleValue()), n_);
        return of(t, DeviceVector::v2s0(FMHIP_OP_CAP, vec_.handle(), vecOf(rv).h), n_);
    }
    RV floor(const RV& rv) const override {
        if (rv->getTypePriority() > priority_) return rv->floor(self());
        const double t = std::max(time_, rv->getFiltrationTime());
        if (isDeterministic() && rv->isDeterministic()) return of(t, jmax(value_, rv->doubleValue()));
        if (isDeterministic()) return of(t, DeviceVector::v1s1(FMHIP_OP_FLOOR_S, vecOf(rv).h, value_), rv->size());
        if (rv->isDeterministic()) return of(t, DeviceVector::v1s1(FMHIP_OP_FLOOR_S, vec_.handle(), rv->doubleValue()), n_);
        return of(t, DeviceVector::v2s0(FMHIP_OP_FLOOR, vec_.handle(), vecOf(rv).h), n_);
    }
    RV accrue(const RV& rate, double p) const override {                            // :1583-1601
        if (rate->getTypePriority() > priority_) return rate->mult(p)->add(1.0)->mult(self());
        const double t = std::max(time_, rate->getFiltrationTime());
        if (rate->isDeterministic()) return mult(1.0 + rate->doubleValue() * p);
        if (isDeterministic()) { const RV r = rate->mult(p)->add(1.0)->mult(value_); return of(t, ownedVec(r), r->size()); }   // newTime kept (twin:1214-1219)
        return of(t, DeviceVector::v2s1(FMHIP_OP_ACCRUE, vec_.handle(), vecOf(rate).h, p), n_);
    }
    RV discount(const RV& rate, double p) const override {                          // :1604-1624
        if (rate->getTypePriority() > priority_) return rate->mult(p)->add(1.0)->invert()->mult(self());
        const double t = std::max(time_, rate->getFiltrationTime());
        if (rate->isDeterministic()) return div(1.0 + rate->doubleValue() * p);
        if (isDeterministic()) { const RV r = rate->mult(p)->add(1.0)->vid(value_); return of(t, ownedVec(r), r->size()); }    // twin:1242-1247 (no zero short-cut)
        return of(t, DeviceVector::v2s1(FMHIP_OP_DISCOUNT, vec_.handle(), vecOf(rate).h, p), n_);
    }
    RV choose(const RV& a, const RV& b) const override {                            // twin:1264-1285
        const double t = std::max(std::max(time_, a->getFiltrationTime()), b->getFiltrationTime());
        if (isDeterministic()) return value_ >= 0 ? a : b;
        const DeviceVector va = a->isDeterministic() ? DeviceVector::filled(n_, a->doubleValue()) : ownedVec(a);
        const DeviceVector vb = b->isDeterministic() ? DeviceVector::filled(n_, b->doubleValue()) : ownedVec(b);
        return of(t, DeviceVector::v3s0(FMHIP_OP_CHOOSE, vec_.handle(), va.handle(), vb.handle()), n_);
    }
    // :1686-1695 compose add/sub(div); the filtration time is the maximum of all three, as in the twin (twin:1395-1438)
    RV addRatio(const RV& num, const RV& den) const override { return restamp(add(num->div(den)), std::max(std::max(time_, num->getFiltrationTime()), den->getFiltrationTime())); }
    RV subRatio(const RV& num, const RV& den) const override { return restamp(sub(num->div(den)), std::max(std::max(time_, num->getFiltrationTime()), den->getFiltrationTime())); }
    RV addProduct(const RV& f1, double f2) const override {                         // :1638-1656
        if (f1->getTypePriority() > priority_) return f1->mult(f2)->add(self());
        const double t = std::max(time_, f1->getFiltrationTime());
        if (f1->isDeterministic()) return add(f1->doubleValue() * f2);
        if (!isDeterministic()) return of(t, DeviceVector::v2s1(FMHIP_OP_ADDPRODUCT_VS, vec_.handle(), vecOf(f1).h, f2), n_);
        return add(f1->mult(f2));
    }
    RV addProduct(const RV& f1, const RV& f2) const override {                      // :1658-1683
        if (f1->getTypePriority() > priority_ || f2->getTypePriority() > priority_) return f1->mult(f2)->add(self());
        const double t = std::max(std::max(time_, f1->getFiltrationTime()), f2->getFiltrationTime());
        if (isDeterministic() && f1->isDeterministic() && f2->isDeterministic()) return of(t, value_ + f1->doubleValue() * f2->doubleValue());
        if (f1->isDeterministic() && f2->isDeterministic()) return add(f1->doubleValue() * f2->doubleValue());
        if (f2->isDeterministic()) return addProduct(f1, f2->doubleValue());
        if (f1->isDeterministic()) return addProduct(f2, f1->doubleValue());
        if (!isDeterministic()) return of(t, DeviceVector::v3s0(FMHIP_OP_ADDPRODUCT, vec_.handle(), vecOf(f1).h, vecOf(f2).h), n_);
        return add(f1->mult(f2));
    }

private:
    // getRandomVariableCuda(rv).realizations (:759-766): foreign types are uploaded through getRealizations()
    static DeviceVector ownedVec(const RV& rv) {                 // an owning reference (shared through the engine's count) or an upload
        if (auto h = dynamic_cast<const RandomVariableHip*>(rv.get())) return h->vec_.share();
        return DeviceVector::fromHost(rv->getRealizations());
    }
    struct VecRef { fmhip_vec h; DeviceVector uploaded; };       // a borrowed handle, or an upload that lives as long as the reference
    static VecRef vecOf(const RV& rv) {
        if (auto h = dynamic_cast<const RandomVariableHip*>(rv.get())) return { h->vec_.handle(), DeviceVector() };
        DeviceVector up = DeviceVector::fromHost(rv->getRealizations());
        const fmhip_vec handle = up.handle();
        return { handle, std::move(up) };
    }
    RV scalar(int op, double s, double detResult) const {
        if (isDeterministic()) return of(time_, detResult);
        return of(time_, DeviceVector::v1s1(op, vec_.handle(), s), n_);
    }
    RV unary(int op, double detResult) const {
        if (isDeterministic()) return of(time_, detResult);
        return of(time_, DeviceVector::v1s0(op, vec_.handle()), n_);
    }
    double time_;
    double value_;
    DeviceVector vec_;
    int64_t n_;
    int priority_;
};

inline double RandomVariable::getQuantile(double quantile) const {
    if (isDeterministic()) return doubleValue();
    const int64_t n = size();
    if (n == 0) return std::nan("");
    std::vector<double> v = getRealizations();
    std::sort(v.begin(), v.end());
    const int64_t idx = (int64_t)std::floor((double)(n + 1) * (1.0 - quantile) - 1.0 + 0.5);
    return v[(size_t)std::min(std::max<int64_t>(idx, 0), n - 1)];
}
inline double RandomVariable::getQuantileExpectation(double quantileStart, double quantileEnd) const {
    if (isDeterministic()) return doubleValue();
    const int64_t n = size();
    if (n == 0) return std::nan("");
    if (quantileStart > quantileEnd) return getQuantileExpectation(quantileEnd, quantileStart);
    std::vector<double> v = getRealizations();
    std::sort(v.begin(), v.end());
    auto index = [n](double q) { return std::min(std::max<int64_t>((int64_t)std::floor((double)(n + 1) * q - 1.0 + 0.5), 0), n - 1); };
    const int64_t i0 = index(quantileStart), i1 = index(quantileEnd);
    double sum = 0.0;
    for (int64_t i = i0; i <= i1; ++i) sum += v[(size_t)i];
    return sum / (double)(i1 - i0 + 1);
}
inline std::vector<double> RandomVariable::getHistogram(const std::vector<double>& intervalPoints) const {
    std::vector<double> hist(intervalPoints.size() + 1, 0.0);
    if (isDeterministic()) {
        const double value = doubleValue();
        for (size_t k = 0; k < intervalPoints.size(); ++k) if (value > intervalPoints[k]) { hist[k] = 1.0; break; }
        hist[intervalPoints.size()] = 1.0;
        return hist;
    }
    std::vector<double> v = getRealizations();
    std::sort(v.begin(), v.end());
    size_t prev = 0;
    for (size_t k = 0; k < intervalPoints.size(); ++k) {
        const size_t cur = std::max<size_t>((size_t)(std::upper_bound(v.begin(), v.end(), intervalPoints[k]) - v.begin()), prev);
        hist[k] = (double)(cur - prev);
        prev = cur;
    }
    hist[intervalPoints.size()] = (double)(v.size() - prev);
    if (!v.empty()) for (double& h : hist) h /= (double)v.size();
    return hist;
}
inline std::vector<std::vector<double>> RandomVariable::getHistogram(int numberOfPoints, double standardDeviations) const {
    const double center = getAverage(), radius = standardDeviations * getStandardDeviation(), step = (numberOfPoints - 1) / 2.0;
    std::vector<double> points((size_t)numberOfPoints), anchors((size_t)numberOfPoints + 1);
    for (int i = 0; i < numberOfPoints; ++i) {
        const double alpha = (-(numberOfPoints - 1) / 2.0 + i) / step;
        points[(size_t)i] = center + alpha * radius;
        anchors[(size_t)i] = center + alpha * radius - radius / (2 * step);
    }
    anchors[(size_t)numberOfPoints] = center + radius + radius / (2 * step);
    return { anchors, getHistogram(points) };
}
inline RV RandomVariable::average() const { return RandomVariableHip::of(-std::numeric_limits<double>::infinity(), getAverage()); }

// All expectations of one objective evaluation in ONE launch + ONE read-back (extension beyond the interface: the
// interface's getAverage() returns a double immediately, i.e. one synchronisation per product).  Falls back to
// getAverage() for values that are not stochastic RandomVariableHip objects.
inline std::vector<double> getAverages(const std::vector<RV>& values) {
    std::vector<double> out(values.size());
    std::vector<fmhip_vec> handles; std::vector<size_t> where;
    int64_t n = -1; bool uniform = true;
    for (size_t k = 0; k < values.size(); ++k) {
        auto h = dynamic_cast<const RandomVariableHip*>(values[k].get());
        if (h && !h->isDeterministic() && h->size() > 0) { if (n < 0) n = h->size(); uniform &= (h->size() == n); handles.push_back(h->deviceVector().handle()); where.push_back(k); }
        else out[k] = values[k]->getAverage();
    }
    if (!handles.empty() && uniform) {
        std::vector<fmhip_moments> m(handles.size());
        DeviceVector::guarded([&] { return fmhip_reduce_moments_batch(handles.data(), (int)handles.size(), nullptr, m.data()); });
        int world = 1; check(fmhip_expectation_world(&world, nullptr));
        for (size_t i = 0; i < handles.size(); ++i) out[where[i]] = m[i].sum / ((double)n * world);
    } else for (size_t k : where) out[k] = values[k]->getAverage();
    return out;
}

class RandomVariableHipFactory final : public RandomVariableFactory {      // RandomVariableCudaFactory.java:27-34
public:
    RV createRandomVariable(double value) const override { return RandomVariableHip::of(-std::numeric_limits<double>::infinity(), value); }
    RV createRandomVariable(double time, double value) const override { return RandomVariableHip::of(time, value); }
    RV createRandomVariable(double time, const std::vector<double>& values) const override { return std::make_shared<RandomVariableHip>(time, values); }
};

// ------------------------------------------------------------------ BrownianMotionHip
// BrownianMotionCudaWithRandomVariableCuda.java:78-259: increments N(0, dt) with filtration time t_{i+1}, generated
// eagerly on first access (:123-130) — here by ONE launch for all (step, factor) vectors (fmhip_bm_generate).
class BrownianMotionHip final : public BrownianMotion {
public:
    BrownianMotionHip(TimeDiscretization td, int numberOfFactors, int64_t numberOfPaths, int64_t seed, int64_t pathOffset = 0)
        : td_(std::move(td)), factors_(numberOfFactors), paths_(numberOfPaths), seed_(seed), offset_(pathOffset) {}
    RV getBrownianIncrement(int timeIndex, int factor) const override {
        std::call_once(generated_, [this] { generate(); });       // (threads that simulate side by side share one Brownian motion)
        return inc_.at((size_t)timeIndex * factors_ + factor);
    }
    // Time-step grouping on the caller's behalf is the engine's business since round 3 (fmhip_set_step_grouping: it watches for the
    // first use of an increment with a new time index itself, whatever class hands the increments out); process-wide.
    static void setGroupSteps(int steps) { check(fmhip_set_step_grouping(steps, nullptr)); }
    const TimeDiscretization& getTimeDiscretization() const override { return td_; }
    int getNumberOfFactors() const override { return factors_; }
    int64_t getNumberOfPaths() const override { return paths_; }
    int64_t getSeed() const { return seed_; }
    RV getRandomVariableForConstant(double value) const override { return RandomVariableHip::of(-std::numeric_limits<double>::infinity(), value); }
    // :131-139 — new, independent generators (nothing is shared: increments are regenerated on first access)
    std::shared_ptr<BrownianMotionHip> getCloneWithModifiedSeed(int64_t seed) const { return std::make_shared<BrownianMotionHip>(td_, factors_, paths_, seed, offset_); }
    std::shared_ptr<BrownianMotionHip> getCloneWithModifiedTimeDiscretization(const TimeDiscretization& td) const { return std::make_shared<BrownianMotionHip>(td, factors_, paths_, seed_, offset_); }
    bool operator==(const BrownianMotionHip& o) const {            // :230-259: same discretisation, factors, paths and seed
        return factors_ == o.factors_ && paths_ == o.paths_ && seed_ == o.seed_ && offset_ == o.offset_ && td_.asVector() == o.td_.asVector();
    }
private:
    void generate() const {
        const int steps = td_.getNumberOfTimeSteps();
        std::vector<double> dt((size_t)steps);
        for (int i = 0; i < steps; ++i) dt[(size_t)i] = td_.getTimeStep(i);
        std::vector<fmhip_vec> h((size_t)steps * factors_);
        check(fmhip_bm_generate(seed_, steps, factors_, paths_, offset_, dt.data(), h.data()));
        inc_.reserve(h.size());
        for (int i = 0; i < steps; ++i)
            for (int f = 0; f < factors_; ++f)
                inc_.push_back(RandomVariableHip::of(td_.getTime(i + 1), DeviceVector(h[(size_t)i * factors_ + f]), paths_));
    }
    TimeDiscretization td_;
    int factors_;
    int64_t paths_, seed_, offset_;
    mutable std::vector<RV> inc_;
    mutable std::once_flag generated_;
};

// ------------------------------------------------------------------ BrownianMotionFromMersenneRandomNumbers
// finmath-lib's CPU generator (net.finmath.montecarlo.BrownianMotionFromMersenneRandomNumbers, not vendored; host/mersenne.hpp)
// behind the BrownianMotion interface: the increments are drawn on the host and handed to the injected factory, so every
// back end sees the same numbers (…ATMTest.java:283 passes the device factory exactly like this).
class BrownianMotionFromMersenneRandomNumbers final : public BrownianMotion {
public:
    BrownianMotionFromMersenneRandomNumbers(TimeDiscretization td, int numberOfFactors, int64_t numberOfPaths, int seed, const RandomVariableFactory* factory)
        : td_(std::move(td)), factors_(numberOfFactors), paths_(numberOfPaths), seed_(seed), factory_(factory) {}
    RV getBrownianIncrement(int timeIndex, int factor) const override {
        std::call_once(generated_, [this] { generate(); });       // (threads that simulate side by side share one Brownian motion)
        return inc_.at((size_t)timeIndex * factors_ + factor);
    }
    const TimeDiscretization& getTimeDiscretization() const override { return td_; }
    int getNumberOfFactors() const override { return factors_; }
    int64_t getNumberOfPaths() const override { return paths_; }
    int getSeed() const { return seed_; }
    RV getRandomVariableForConstant(double value) const override { return factory_->createRandomVariable(value); }
private:
    void generate() const {
        const int steps = td_.getNumberOfTimeSteps();
        std::vector<double> dt((size_t)steps);
        for (int i = 0; i < steps; ++i) dt[(size_t)i] = td_.getTimeStep(i);
        std::vector<double> all((size_t)steps * factors_ * (size_t)paths_);
        mersenneIncrements(seed_, steps, factors_, paths_, dt.data(), all.data());
        inc_.reserve((size_t)steps * factors_);
        for (int i = 0; i < steps; ++i)
            for (int f = 0; f < factors_; ++f) {
                const double* p = all.data() + ((size_t)i * factors_ + f) * (size_t)paths_;
                inc_.push_back(factory_->createRandomVariable(td_.getTime(i + 1), std::vector<double>(p, p + paths_)));
            }
    }
    TimeDiscretization td_;
    int factors_;
    int64_t paths_;
    int seed_;
    const RandomVariableFactory* factory_;
    mutable std::vector<RV> inc_;
    mutable std::once_flag generated_;
};

} // namespace fmhost
