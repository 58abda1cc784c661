// recall.h - forwards to the host mirror (reference header of the same name: src/recall.h)
#pragma once
#include "gpann_host.h"
