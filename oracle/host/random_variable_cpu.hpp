// random_variable_cpu.hpp — CPU implementation of the fmhost interfaces over the C oracle (TEST INFRASTRUCTURE ONLY:
// used by tests and by the cpu_baseline leg of the benchmarks; never linked into the product).
//
// RandomVariableFromFloatArray restates /root/reference/src/main/java/net/finmath/cuda/cpu/montecarlo/
// RandomVariableFromFloatArray.java at class level (dispatch, time propagation; `:line` cites that file); the
// array arithmetic is oracle/rv_float.c.  Cost model kept: one single-threaded loop and one fresh array per method
// call (:787-791).  BrownianMotionCpu produces the SAME increments as BrownianMotionHip (oracle/philox_normal.c is the
// normative definition of the generator), so a model can be run on both back ends path for path.
#pragma once
#include <algorithm>
#include <limits>
#include "../../finmath-lib-cuda-extensions_amd/host/random_variable.hpp"
#include "../fm_oracle.h"

namespace fmhost {

class RandomVariableFromFloatArray final : public RandomVariable {
public:
    static constexpr int typePriorityDefault = 1;                                   // :47
    using Arr = std::shared_ptr<const std::vector<float>>;
    RandomVariableFromFloatArray(double time, double value) : time_(time), value_(value) {}
    RandomVariableFromFloatArray(double time, Arr a) : time_(time), value_(std::numeric_limits<double>::quiet_NaN()), a_(std::move(a)) {}
    static RV of(double t, double v) { return std::make_shared<RandomVariableFromFloatArray>(t, v); }
    static RV of(double t, Arr a) { return std::make_shared<RandomVariableFromFloatArray>(t, std::move(a)); }
    static RV fromDouble(double t, const std::vector<double>& v) {                   // :177-179, :217-223
        auto a = std::make_shared<std::vector<float>>(v.size());
        orc_f_from_double(v.data(), (int64_t)v.size(), a->data());
        return of(t, a);
    }
    double getFiltrationTime() const override { return time_; }
    int getTypePriority() const override { return typePriorityDefault; }
    bool isDeterministic() const override { return !a_; }
    int64_t size() const override { return a_ ? (int64_t)a_->size() : 1; }
    double doubleValue() const override { if (!a_) return value_; throw UnsupportedOperation("The random variable is non-deterministic"); }
    std::vector<double> getRealizations() const override {
        if (!a_) return { value_ };
        std::vector<double> o(a_->size()); orc_f_to_double(a_->data(), (int64_t)a_->size(), o.data()); return o;
    }
    const Arr& array() const { return a_; }
    double getAverage() const override { return !a_ ? value_ : (a_->empty() ? std::nan("") : orc_f_average(a_->data(), (int64_t)a_->size())); }   // :314
    double getVariance() const override { return (!a_ || a_->size() == 1) ? 0.0 : (a_->empty() ? std::nan("") : orc_f_variance(a_->data(), (int64_t)a_->size())); }   // :360
    double getMin() const override { return !a_ ? value_ : orc_f_min(a_->data(), (int64_t)a_->size()); }
    double getMax() const override { return !a_ ? value_ : orc_f_max(a_->data(), (int64_t)a_->size()); }

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
    // RandomVariableFromFloatArray.java:1395-1438: maximum of all three filtration times
    RV addRatio(const RV& num, const RV& den) const override { return restamp(add(num->div(den)), std::max(std::max(time_, num->getFiltrationTime()), den->getFiltrationTime())); }
    RV subRatio(const RV& num, const RV& den) const override { return restamp(sub(num->div(den)), std::max(std::max(time_, num->getFiltrationTime()), den->getFiltrationTime())); }
    static RV restamp(const RV& rv, double time) {
        const auto* c = dynamic_cast<const RandomVariableFromFloatArray*>(rv.get());
        if (!c || c->time_ == time) return rv;
        return c->a_ ? of(time, c->a_) : of(time, c->value_);
    }
    RV isNaN() const override   { return unary(FMHIP_OP_ISNAN, value_ != value_ ? 1.0 : 0.0); }
    RV sin() const override     { return unary(FMHIP_OP_SIN, std::sin(value_)); }
    RV cos() const override     { return unary(FMHIP_OP_COS, std::cos(value_)); }
    RV apply(const std::function<double(double)>& f) const override {       // RandomVariableFromFloatArray.java:667-676
        if (!a_) return of(time_, f(value_));
        auto out = std::make_shared<std::vector<float>>(a_->size());
        for (size_t i = 0; i < a_->size(); ++i) (*out)[i] = (float)f((double)(*a_)[i]);
        return of(time_, out);
    }

    RV add(const RV& rv) const override { return binary(rv, [&] { return rv->add(self()); }, value_ + det(rv), FMHIP_OP_ADD, FMHIP_OP_ADD_S, false); }    // :961
    RV sub(const RV& rv) const override { return binary(rv, [&] { return rv->bus(self()); }, value_ - det(rv), FMHIP_OP_SUB, FMHIP_OP_BUS_S, false); }    // :990
    RV bus(const RV& rv) const override {                                                                                                           // :1020
        if (rv->getTypePriority() > getTypePriority()) return rv->sub(self());
        const double t = std::max(time_, rv->getFiltrationTime());
        if (!a_ && rv->isDeterministic()) return of(t, rv->doubleValue() - value_);
        if (!a_) return of(t, v1s1(FMHIP_OP_SUB_S, arr(rv, 0), value_));
        return of(t, v2s0(FMHIP_OP_SUB, arr(rv, size()), a_));
    }
    RV mult(const RV& rv) const override {                                                                                                          // :1050
        if (rv->getTypePriority() > getTypePriority()) return rv->mult(self());
        const double t = std::max(time_, rv->getFiltrationTime());
        if (!a_ && rv->isDeterministic()) return of(t, value_ * rv->doubleValue());
        if (rv->isDeterministic()) return mult(rv->doubleValue());
        if (!a_) return of(t, v1s1(FMHIP_OP_MULT_S, arr(rv, 0), value_));
        return of(t, v2s0(FMHIP_OP_MULT, a_, arr(rv, size())));
    }
    RV div(const RV& rv) const override {                                                                                                           // :1082
        if (rv->getTypePriority() > getTypePriority()) return rv->vid(self());
        const double t = std::max(time_, rv->getFiltrationTime());
        if (!a_ && rv->isDeterministic()) return of(t, value_ / rv->doubleValue());
        if (rv->isDeterministic()) return div(rv->doubleValue());
        if (!a_) return of(t, v1s1(FMHIP_OP_VID_S, arr(rv, 0), value_));
        return of(t, v2s0(FMHIP_OP_DIV, a_, arr(rv, size())));
    }
    RV vid(const RV& rv) const override {                                                                                                           // :1115
        if (rv->getTypePriority() > getTypePriority()) return rv->div(self());
        const double t = std::max(time_, rv->getFiltrationTime());
        if (!a_ && rv->isDeterministic()) return of(t, rv->doubleValue() / value_);
        if (!a_) return of(t, v1s1(FMHIP_OP_DIV_S, arr(rv, 0), value_));
        if (rv->isDeterministic()) {             // :1138 (float)(randomVariable.get(i) / realizations[i]): the DOUBLE constant is divided
            auto out = std::make_shared<std::vector<float>>(a_->size());
            const double v = rv->doubleValue();
            for (size_t i = 0; i < a_->size(); ++i) (*out)[i] = (float)(v / (double)(*a_)[i]);
            return of(t, out);
        }
        return of(t, v2s0(FMHIP_OP_DIV, arr(rv, size()), a_));
    }
    RV cap(const RV& rv) const override { return binary(rv, [&] { return rv->cap(self()); }, jmin(value_, det(rv)), FMHIP_OP_CAP, FMHIP_OP_CAP_S, false); }        // :1145
    RV floor(const RV& rv) const override { return binary(rv, [&] { return rv->floor(self()); }, jmax(value_, det(rv)), FMHIP_OP_FLOOR, FMHIP_OP_FLOOR_S, false); }  // :1174
    RV accrue(const RV& rate, double p) const override {                                                                                            // :1203
        if (rate->getTypePriority() > getTypePriority()) return rate->mult(p)->add(1.0)->mult(self());
        const double t = std::max(time_, rate->getFiltrationTime());
        if (rate->isDeterministic()) return mult(1.0 + rate->doubleValue() * p);
        if (!a_) return of(t, v1s1(FMHIP_OP_MULT_S, v1s1(FMHIP_OP_ADD_S, v1s1(FMHIP_OP_MULT_S, arr(rate, 0), p), 1.0), value_));
        return of(t, v2s1(FMHIP_OP_ACCRUE, a_, arr(rate, size()), p));
    }
    RV discount(const RV& rate, double p) const override {                                                                                          // :1231
        if (rate->getTypePriority() > getTypePriority()) return rate->mult(p)->add(1.0)->vid(self());
        const double t = std::max(time_, rate->getFiltrationTime());
        if (rate->isDeterministic()) return div(1.0 + rate->doubleValue() * p);
        if (!a_) return of(t, v1s1(FMHIP_OP_VID_S, v1s1(FMHIP_OP_ADD_S, v1s1(FMHIP_OP_MULT_S, arr(rate, 0), p), 1.0), value_));
        return of(t, v2s1(FMHIP_OP_DISCOUNT, a_, arr(rate, size()), p));
    }
    RV choose(const RV& a, const RV& b) const override {                                                                                            // :1264
        const double t = std::max(std::max(time_, a->getFiltrationTime()), b->getFiltrationTime());
        if (!a_) return value_ >= 0 ? a : b;
        return of(t, v3s0(FMHIP_OP_CHOOSE, a_, arr(a, size()), arr(b, size())));
    }
    RV addProduct(const RV& f1, double f2) const override {                                                                                         // :1318
        if (f1->getTypePriority() > getTypePriority()) return f1->mult(f2)->add(self());
        const double t = std::max(time_, f1->getFiltrationTime());
        if (f1->isDeterministic()) return add(f1->doubleValue() * f2);
        if (!a_) return of(t, v1s1(FMHIP_OP_ADD_S, v1s1(FMHIP_OP_MULT_S, arr(f1, 0), f2), value_));
        return of(t, v2s1(FMHIP_OP_ADDPRODUCT_VS, a_, arr(f1, size()), f2));
    }
    RV addProduct(const RV& f1, const RV& f2) const override {                                                                                      // :1354
        if (f1->getTypePriority() > getTypePriority() || f2->getTypePriority() > getTypePriority()) return f1->mult(f2)->add(self());
        const double t = std::max(std::max(time_, f1->getFiltrationTime()), f2->getFiltrationTime());
        if (!a_ && f1->isDeterministic() && f2->isDeterministic()) return of(t, value_ + f1->doubleValue() * f2->doubleValue());
        if (f1->isDeterministic() && f2->isDeterministic()) return add(f1->doubleValue() * f2->doubleValue());
        if (f2->isDeterministic()) return addProduct(f1, f2->doubleValue());
        if (f1->isDeterministic()) return addProduct(f2, f1->doubleValue());
        if (a_) return of(t, v3s0(FMHIP_OP_ADDPRODUCT, a_, arr(f1, size()), arr(f2, size())));
        return add(f1->mult(f2));
    }

private:
    static double det(const RV& rv) { return rv->isDeterministic() ? rv->doubleValue() : std::numeric_limits<double>::quiet_NaN(); }
    // float view of an operand as the Java loops see it: realizations[i] or (float)value
    static Arr arr(const RV& rv, int64_t n) {
        if (auto c = dynamic_cast<const RandomVariableFromFloatArray*>(rv.get())) {
            if (c->a_) return c->a_;
            return std::make_shared<std::vector<float>>((size_t)n, (float)c->value_);
        }
        if (rv->isDeterministic()) return std::make_shared<std::vector<float>>((size_t)n, (float)rv->doubleValue());
        const std::vector<double> d = rv->getRealizations();
        auto a = std::make_shared<std::vector<float>>(d.size());
        orc_f_from_double(d.data(), (int64_t)d.size(), a->data());
        return a;
    }
    static Arr v1s0(int op, const Arr& a) { auto o = std::make_shared<std::vector<float>>(a->size()); orc_f_v1s0(op, a->data(), (int64_t)a->size(), o->data()); return o; }
    static Arr v1s1(int op, const Arr& a, double s) { auto o = std::make_shared<std::vector<float>>(a->size()); orc_f_v1s1(op, a->data(), s, (int64_t)a->size(), o->data()); return o; }
    static Arr v2s0(int op, const Arr& a, const Arr& b) { auto o = std::make_shared<std::vector<float>>(a->size()); orc_f_v2s0(op, a->data(), b->data(), (int64_t)a->size(), o->data()); return o; }
    static Arr v2s1(int op, const Arr& a, const Arr& b, double s) { auto o = std::make_shared<std::vector<float>>(a->size()); orc_f_v2s1(op, a->data(), b->data(), s, (int64_t)a->size(), o->data()); return o; }
    static Arr v3s0(int op, const Arr& a, const Arr& b, const Arr& c) { auto o = std::make_shared<std::vector<float>>(a->size()); orc_f_v3s0(op, a->data(), b->data(), c->data(), (int64_t)a->size(), o->data()); return o; }
    RV scalar(int op, double s, double detResult) const { return a_ ? of(time_, v1s1(op, a_, s)) : of(time_, detResult); }
    RV unary(int op, double detResult) const { return a_ ? of(time_, v1s0(op, a_)) : of(time_, detResult); }
    template <class Swapped>
    RV binary(const RV& rv, Swapped swapped, double detResult, int opVV, int opDetReceiver, bool) const {
        if (rv->getTypePriority() > getTypePriority()) return swapped();
        const double t = std::max(time_, rv->getFiltrationTime());
        if (!a_ && rv->isDeterministic()) return of(t, detResult);
        if (!a_) return of(t, v1s1(opDetReceiver, arr(rv, 0), value_));
        return of(t, v2s0(opVV, a_, arr(rv, size())));
    }
    double time_, value_;
    Arr a_;
};

class RandomVariableFloatFactory final : public RandomVariableFactory {          // RandomVariableFloatFactory.java:24-35
public:
    RV createRandomVariable(double value) const override { return RandomVariableFromFloatArray::of(-std::numeric_limits<double>::infinity(), value); }
    RV createRandomVariable(double time, double value) const override { return RandomVariableFromFloatArray::of(time, value); }
    RV createRandomVariable(double time, const std::vector<double>& values) const override { return RandomVariableFromFloatArray::fromDouble(time, values); }
};

class BrownianMotionCpu final : public BrownianMotion {
public:
    BrownianMotionCpu(TimeDiscretization td, int factors, int64_t paths, int64_t seed, int64_t pathOffset = 0)
        : td_(std::move(td)), factors_(factors), paths_(paths), seed_(seed), offset_(pathOffset) {}
    RV getBrownianIncrement(int timeIndex, int factor) const override {
        if (inc_.empty()) {
            const int steps = td_.getNumberOfTimeSteps();
            for (int i = 0; i < steps; ++i)
                for (int f = 0; f < factors_; ++f) {
                    auto a = std::make_shared<std::vector<float>>((size_t)paths_);
                    orc_bm_increment(seed_, (uint32_t)(i * factors_ + f), offset_, paths_, (float)std::sqrt(td_.getTimeStep(i)), a->data());
                    inc_.push_back(RandomVariableFromFloatArray::of(td_.getTime(i + 1), RandomVariableFromFloatArray::Arr(a)));
                }
        }
        return inc_.at((size_t)timeIndex * factors_ + factor);
    }
    const TimeDiscretization& getTimeDiscretization() const override { return td_; }
    int getNumberOfFactors() const override { return factors_; }
    int64_t getNumberOfPaths() const override { return paths_; }
    RV getRandomVariableForConstant(double value) const override { return RandomVariableFromFloatArray::of(-std::numeric_limits<double>::infinity(), value); }
private:
    TimeDiscretization td_;
    int factors_;
    int64_t paths_, seed_, offset_;
    mutable std::vector<RV> inc_;
};

} // namespace fmhost
