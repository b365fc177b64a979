// TEST INFRASTRUCTURE -- oracle/_ref recipe: stands in for the un-vendored header of the same name (see vk_stub.h)
#pragma once
#include "../vk_stub.h"
