// dist.h - forwards to the host mirror (reference header of the same name: src/dist.h)
#pragma once
#include "gpann_host.h"
