// tests/cpp/test_bfgsmat_class.cpp -- a program that drives the limited-memory matrix ITSELF, through the reference's class
// name and member signatures (/root/reference/include/LBFGSpp/BFGSMat.h:61 reset, :81 add_correction, :276 apply_Hv, :307
// theta, :310 num_corrections, :150/:211 get_Bmat / get_Hmat) with Eigen vectors (the stand-in under oracle/eigen_shim; real
// Eigen is not installed here) and with std::vector.
//   test_bfgsmat_class n m npairs f64|f32 out.bin
// feeds npairs deterministic pairs through add_correction (more than m: the ring wraps), applies a * H to a vector and
// writes S, Y, v, the product, theta and the number of corrections; tests/test_dropin_gpu.py compares them with the reference
// build's own BFGSMat (oracle/_ref) on the same numbers.
#include <Eigen/Core>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <vector>

#include <LBFGSpp/BFGSMat.h>

using namespace LBFGSpp;

static std::uint64_t g_state = 0x9E3779B97F4A7C15ull;
static double u01()
{
    g_state = g_state * 6364136223846793005ull + 1442695040888963407ull;
    return double(g_state >> 11) * (1.0 / 9007199254740992.0);
}

template <typename Scalar>
static int run(int n, int m, int npairs, const char* path)
{
    typedef Eigen::Matrix<Scalar, Eigen::Dynamic, 1> Vector;
    std::FILE* fp = std::fopen(path, "wb");
    if (!fp)
        return 2;
    BFGSMat<Scalar> bfgs;
    bool threw = false;
    try
    {
        (void) bfgs.theta();  // before reset(): refused, not a crash
    }
    catch (const std::logic_error&)
    {
        threw = true;
    }
    if (!threw)
        return 3;
    bfgs.reset(n, m);
    if (bfgs.num_corrections() != 0 || bfgs.theta() != Scalar(1))
        return 4;
    Vector s(n), y(n);
    for (int k = 0; k < npairs; k++)
    {
        for (int i = 0; i < n; i++)
        {
            s[i] = Scalar(2.0 * u01() - 1.0);
            y[i] = Scalar(double(s[i]) * (1.0 + u01()) + 0.05 * (2.0 * u01() - 1.0));  // s.y > 0 like a curvature pair
        }
        bfgs.add_correction(s, y);
        std::fwrite(s.data(), sizeof(Scalar), size_t(n), fp);
        std::fwrite(y.data(), sizeof(Scalar), size_t(n), fp);
    }
    std::vector<Scalar> v(static_cast<size_t>(n)), res;  // the other vector type a caller may hold
    for (int i = 0; i < n; i++)
        v[size_t(i)] = Scalar(2.0 * u01() - 1.0);
    bfgs.apply_Hv(v, Scalar(-1), res);
    if (std::int64_t(res.size()) != n)
        return 5;
    Vector res2;
    Vector v2(n);
    for (int i = 0; i < n; i++)
        v2[i] = v[size_t(i)];
    bfgs.apply_Hv(v2, Scalar(-1), res2);  // again, Eigen types: the same bits
    for (int i = 0; i < n; i++)
        if (std::memcmp(&res2[i], &res[size_t(i)], sizeof(Scalar)) != 0)
            return 6;
    std::fwrite(v.data(), sizeof(Scalar), size_t(n), fp);
    std::fwrite(res.data(), sizeof(Scalar), size_t(n), fp);
    const double tail[2] = {double(bfgs.theta()), double(bfgs.num_corrections())};
    std::fwrite(tail, sizeof(double), 2, fp);
    if (n <= 64)
    {
        // the dense getters agree with the product: H v from get_Hmat against apply_Hv, and B H = I
        const auto H = bfgs.get_Hmat();
        const auto B = bfgs.get_Bmat();
        double worst = 0, worst_id = 0, scale = 1e-300;
        for (int i = 0; i < n; i++)
        {
            double acc = 0;
            for (int j = 0; j < n; j++)
                acc += double(H(i, j)) * double(v[size_t(j)]);
            worst = std::fmax(worst, std::fabs(-acc - double(res[size_t(i)])));
            scale = std::fmax(scale, std::fabs(double(res[size_t(i)])));
            for (int j = 0; j < n; j++)
            {
                double bh = 0;
                for (int k = 0; k < n; k++)
                    bh += double(B(i, k)) * double(H(k, j));
                worst_id = std::fmax(worst_id, std::fabs(bh - (i == j ? 1.0 : 0.0)));
            }
        }
        const double tol = sizeof(Scalar) == 8 ? 1e-9 : 2e-2;
        if (!(worst <= tol * scale) || !(worst_id <= tol))
        {
            std::printf("dense getters: |H v - apply_Hv| %.3e (scale %.3e), |B H - I| %.3e\n", worst, scale, worst_id);
            return 7;
        }
    }
    // a wrong-sized vector is refused
    threw = false;
    try
    {
        Vector bad(n + 1);
        bfgs.add_correction(bad, y);
    }
    catch (const std::invalid_argument&)
    {
        threw = true;
    }
    std::fclose(fp);
    return threw ? 0 : 8;
}

int main(int argc, char** argv)
{
    if (argc != 6)
        return 1;
    const int n = std::atoi(argv[1]), m = std::atoi(argv[2]), npairs = std::atoi(argv[3]);
    int rc;
    try
    {
        rc = std::strcmp(argv[4], "f32") == 0 ? run<float>(n, m, npairs, argv[5]) : run<double>(n, m, npairs, argv[5]);
    }
    catch (const std::exception& e)
    {
        std::printf("exception: %s\n", e.what());
        return 9;
    }
    std::printf(rc == 0 ? "BFGSMAT OK\n" : "BFGSMAT FAIL %d\n", rc);
    return rc;
}
