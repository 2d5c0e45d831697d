// Test infrastructure: entry point for the reference's tests/test_predictors.cpp when it is compiled against the
// product's mirror of Block::Encoder (see oracle/Makefile, target ref-tests).  The reference calls this function
// from tests/test_lpc.cpp:185; nothing else of that file is needed here.
#include <cstdio>

#include "lacx.h"

void run_predictor_tests();  // defined by the reference's tests/test_predictors.cpp:64

int main() {
    if (lacx_device_count() < 1) {
        std::printf("no HIP device: built and linked only\n");
        return 77;
    }
    run_predictor_tests();
    return 0;
}
