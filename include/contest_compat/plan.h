// Lets radix-join_amd/host/contest_execute.cpp say `#include <plan.h>` exactly as the
// reference's src/execute.cpp does; inside the reference tree the real header is found
// first, here it forwards to this repository's declaration of the same contract.
#pragma once
#include "../rj_contest_api.hpp"
