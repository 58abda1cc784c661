// routes.h - forwards to the host mirror (reference header of the same name: src/routes.h)
#pragma once
#include "gpann_host.h"
