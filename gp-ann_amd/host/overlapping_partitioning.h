// overlapping_partitioning.h - forwards to the host mirror (reference header of the same name: src/overlapping_partitioning.h)
#pragma once
#include "gpann_host.h"
