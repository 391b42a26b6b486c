// host shim (oracle test infrastructure): see cuda_host_shim.h
#pragma once
#include "cuda_host_shim.h"
