// partitioning.h - forwards to the host mirror (reference header of the same name: src/partitioning.h)
#pragma once
#include "gpann_host.h"
