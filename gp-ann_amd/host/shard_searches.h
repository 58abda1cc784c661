// shard_searches.h - forwards to the host mirror (reference header of the same name: src/shard_searches.h)
#pragma once
#include "gpann_host.h"
