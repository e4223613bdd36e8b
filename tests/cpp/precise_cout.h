// Force-included (-include) when the reference's example programs are compiled for the GPU-vs-reference comparison
// (tests/ref_examples.py): the examples print with the stream's default 6 significant digits; this sets 17 before main()
// runs, so that iterates and objective values can be compared to 1e-10 without editing a line of the examples.
#pragma once
#include <iomanip>
#include <iostream>
namespace {
struct LbfgsxPreciseCout
{
    LbfgsxPreciseCout() { std::cout << std::setprecision(17); }
} lbfgsx_precise_cout_instance;
}  // namespace
