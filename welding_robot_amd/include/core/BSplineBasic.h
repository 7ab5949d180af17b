// BSplineBasic.h -- drop-in for the reference's core/BSplineBasic.h on the C ABI of libweldacs.so
// (wa_bspline_*, include/weldacs.h).  Same class template and public members
// (BS_Basic<T, DIM, DEGREE, CONST_LEVEL_INI, CONST_LEVEL_FIN>: SetParam / getCurvePoint / getCurveDerPoint),
// so main.cpp:299-300, :337-338 compile unchanged; knots, constrained control points and the evaluation
// run in the k_bspline_* HIP kernels and are bit-identical to the reference class (fp32, same operation order).
//
// What a maintainer should know:
//  * T must be float (the only instantiation the reference uses); DIM 1..16, DEGREE 0..7, constraint
//    levels <= DEGREE (above that the reference indexes its work arrays out of bounds);
//  * BS_Basic<float,3,2,2,2> (main.cpp:337) makes the reference read two never-written heap cells
//    (c_mat[idx][CONST_LEVEL_FIN+1], BSplineBasic.h:414-431); that value is explicit here:
//    setUninitializedValue(v), default 0;
//  * getCurvePoint / getCurveDerPoint evaluate ONE time on the host (wa_bspline_eval_host: knots and control points are mirrored
//    once per SetParam; same fp32 operations in the same order as the kernel, bit-identical, ~0.1-0.2 us per call), because
//    main.cpp:302-316 / :341-351 call them inside clock()-paced loops whose sample count depends on the call's duration;
//    setHostEvaluation(false) sends every call through the device instead (a launch + synchronise + copy each);
//  * sample(t0, dt, count, out) evaluates a whole fixed-rate time series in one launch -- the replacement
//    for main.cpp's clock()-paced loops (:302-316, :341-351), whose sample count depends on CPU speed;
//  * the destructor frees the device arrays (the reference leaks Knots_ / CPoints_).
#ifndef B_SPLINE_BASIC
#define B_SPLINE_BASIC
#include <assert.h>
#include <stdio.h>
#include <string.h>

#include <iostream>
#include <type_traits>
#include <vector>

#include "weldacs_dropin.h"

template <typename T, int DIM, int DEGREE, int CONST_LEVEL_INI, int CONST_LEVEL_FIN>
class BS_Basic {
    static_assert(std::is_same<T, float>::value, "libweldacs evaluates BS_Basic in fp32 only");

public:
    BS_Basic(int _NUM_MIDDLE) : h_(NULL), status_(WA_OK), num_middle_(_NUM_MIDDLE), host_eval_(true)
    {
        wa_ctx *ctx = weldacs_dropin::context();
        if (!ctx) { status_ = WA_ERR_DEVICE; return; }
        status_ = wa_bspline_create(ctx, DIM, DEGREE, CONST_LEVEL_INI, CONST_LEVEL_FIN, _NUM_MIDDLE, &h_);
        if (status_ != WA_OK) printf("Invalid setup: %s\n", wa_last_error(ctx));   // reference :53-55
    }
    ~BS_Basic() { wa_bspline_destroy(h_); }
    BS_Basic(const BS_Basic &) = delete;
    BS_Basic &operator=(const BS_Basic &) = delete;

    // reference :72-78.  init / fin: (level+1) x DIM values; middle_pt: NUM_MIDDLE rows, first DIM of each used
    bool SetParam(T *init, T *fin, T **middle_pt, T fin_time)
    {
        if (!h_) return false;
        std::vector<float> mid((size_t)num_middle_ * DIM);
        for (int i = 0; i < num_middle_; ++i) memcpy(&mid[(size_t)i * DIM], middle_pt[i], sizeof(float) * DIM);
        status_ = wa_bspline_set_param(h_, init, fin, mid.data(), DIM, fin_time);
        return status_ == WA_OK;
    }

    // reference :87-112
    bool getCurvePoint(T u, T *ret) { return one(u, 0, ret); }

    // reference :122-146
    bool getCurveDerPoint(T u, int d, T *ret)
    {
        if (d > DEGREE) return false;
        return one(u, d, ret);
    }

    // ---- extensions -------------------------------------------------------------------------
    // u_i = t0 + i*dt (fp32) for i < count, position (der = 0) or der-th derivative; out = count x DIM
    bool sample(T t0, T dt, int count, std::vector<T> &out, int der = 0)
    {
        if (!h_) return false;
        out.assign((size_t)count * DIM, 0.0f);
        status_ = wa_bspline_sample(h_, t0, dt, count, der, out.data(), NULL, NULL);
        return status_ == WA_OK;
    }
    void setUninitializedValue(float v)
    {
        uint32_t b;
        memcpy(&b, &v, 4);
        if (h_) wa_bspline_set_uninit(h_, b);
    }
    void setHostEvaluation(bool on) { host_eval_ = on; }
    int lastStatus() const { return status_; }
    wa_bspline *handle() { return h_; }

private:
    bool one(T u, int d, T *ret)
    {
        if (!h_) return false;
        float out[DIM];
        uint8_t ok = 0;
        status_ = host_eval_ ? wa_bspline_eval_host(h_, u, d, out, &ok) : wa_bspline_eval(h_, &u, 1, d, out, &ok);
        if (status_ != WA_OK || !ok) return false;      // `ret` untouched, as in the reference
        for (int i = 0; i < DIM; ++i) ret[i] = out[i];
        return true;
    }
    wa_bspline *h_;
    int status_;
    int num_middle_;
    bool host_eval_;
};

#endif
