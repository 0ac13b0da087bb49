#pragma once
#include "GpuResources.h"
