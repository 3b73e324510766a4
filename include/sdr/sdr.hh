// sdr.hh — umbrella header of this repository's API-compatible sdr:: core plus the MI355X nodes.
// An application written against the reference's <libsdr/sdr.hh> node API (Sink/Source/Buffer/Queue)
// can include this instead; an application that keeps the reference core includes the reference's
// sdr.hh first and then only "sdr/gpu/nodes.hh" (INTEGRATION.md).
#ifndef SDR_CORE_SDR_HH
#define SDR_CORE_SDR_HH
#include "exception.hh"
#include "logger.hh"
#include "buffer.hh"
#include "queue.hh"
#include "node.hh"
#include "siggen.hh"
#include "utils.hh"
#include "wavfile.hh"
#include "gpu/design.hh"
#include "gpu/nodes.hh"
#endif
