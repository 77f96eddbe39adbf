// ORACLE SUPPORT (test infrastructure; never linked into or loaded by the product).
//
// The MAIN path's pile selection of the reference, src/daccord.cpp:2026-2105 (the heap entry, its block order and the per-thread
// buffers) and :2120-2288 (the body of the loop over A reads up to the sort by abpos), compiled FROM THE REFERENCE'S OWN LINES:
// oracle/ref_shim/build.sh cuts the two ranges out of /root/reference/src/daccord.cpp for the duration of the compile and this
// file includes them.  Around them stand
//   - the libmaus2 stand-in of this directory (shim.hpp: AutoArray, FiniteSizeHeap), and
//   - three stand-ins that only this translation unit needs, defined below: the view of a RAW overlap record (40 byte header +
//     trace), the block parser and the input stream.
// What stays an assumption, because libmaus2's OverlapParser is not part of /root/reference: a record that straddles the end of
// a 64 KiB input block is delivered with the block in which its last byte arrives (parseBlock keeps the incomplete tail and
// completes it in front of the next block's records).  Everything else of the selection -- score, heap order and its ties,
// keep-the-worst eviction, the copy order (last block backwards, earlier blocks forwards), the unstable sort by abpos -- is the
// reference's code.
#include <libmaus2/shim.hpp>
#include <map>
#include <sstream>
#include <string>
#include <vector>
#include <cstring>
#include "../../include/daccord_hip.h"

namespace libmaus2 {
namespace aio {
// name -> contents of the "files" this translation unit serves
static std::map<std::string,std::string> & selRegistry() { static std::map<std::string,std::string> M; return M; }
struct InputStreamInstance : public std::istringstream
{
	typedef ::libmaus2::util::unique_ptr<InputStreamInstance>::type unique_ptr_type;
	InputStreamInstance(std::string const & fn) : std::istringstream(selRegistry()[fn], std::ios::in | std::ios::binary) {}
};
}
namespace dazzler { namespace align {
static inline int32_t selGet32(uint8_t const * p) { int32_t v; std::memcpy(&v,p,4); return v; }
// DALIGNER's record as it lies in a .las file: tlen, diffs, abpos, bbpos, aepos, bepos, flags, aread, bread (9 x int32), 4 bytes
// of padding, then tlen trace values of 1 (tspace <= 128) or 2 bytes
struct RawOverlapDataInterface
{
	uint8_t const * p;
	RawOverlapDataInterface() : p(0) {}
	RawOverlapDataInterface(uint8_t const * rp) : p(rp) {}
	int64_t tlen() const { return selGet32(p+0); }
	int64_t diffs() const { return selGet32(p+4); }
	int64_t abpos() const { return selGet32(p+8); }
	int64_t bbpos() const { return selGet32(p+12); }
	int64_t aepos() const { return selGet32(p+16); }
	int64_t bepos() const { return selGet32(p+20); }
	uint64_t flags() const { return static_cast<uint32_t>(selGet32(p+24)); }
	int64_t aread() const { return selGet32(p+28); }
	int64_t bread() const { return selGet32(p+32); }
};
struct OverlapData
{
	std::vector<uint8_t> D;              // the complete records of the current block, back to back
	std::vector<uint64_t> O;             // their start offsets + the end
	uint64_t size() const { return O.size() ? O.size()-1 : 0; }
	std::pair<uint8_t const *,uint8_t const *> getData(uint64_t const j) const
	{
		return std::pair<uint8_t const *,uint8_t const *>(D.data()+O[j],D.data()+O[j+1]);
	}
};
struct OverlapParser
{
	typedef ::libmaus2::util::unique_ptr<OverlapParser>::type unique_ptr_type;
	enum split_type { overlapparser_do_split, overlapparser_do_not_split };
	uint64_t tbytes;
	std::vector<uint8_t> tail;           // bytes of a record whose end has not arrived yet
	OverlapData data;
	OverlapParser(int64_t const tspace) : tbytes(tspace <= 128 ? 1 : 2) {}
	bool isIdle() const { return tail.empty(); }
	OverlapData & getData() { return data; }
	void parseBlock(uint8_t const * pa, uint8_t const * pe, split_type)
	{
		std::vector<uint8_t> B(tail); B.insert(B.end(),pa,pe); tail.clear();
		data.D.clear(); data.O.clear();
		uint64_t pos = 0;
		while ( true )
		{
			if ( B.size() - pos < 40 ) break;
			uint64_t const s = 40 + static_cast<uint64_t>(selGet32(B.data()+pos))*tbytes;
			if ( B.size() - pos < s ) break;
			data.O.push_back(pos);
			pos += s;
		}
		data.O.push_back(pos);
		if ( data.O.size() == 1 ) data.O.clear();
		tail.assign(B.begin()+pos,B.end());
		data.D.assign(B.begin(),B.begin()+pos);
	}
};
}}}

// from here on the reference's lines see the raw view under the name they use
#define OverlapDataInterface RawOverlapDataInterface

#undef NDEBUG
#include <cassert>

#if defined(DACC_REF_DEFAULTS_EXCERPT)
namespace refdefaults {
#include DACC_REF_DEFAULTS_EXCERPT
}
#endif

extern "C" {

// One pile's records (file order) -> raw .las bytes -> the reference's selection (:2120-2288) -> the selected records in the
// order the reference hands to HandleContext::operator().  vard/rl/avgreadlength feed the reference's own lmaxinput formula
// (:2120-2125; vard = 0 means "maxinput as given").
int ref_pile_select(dacc_overlap const * in, uint64_t n, int trace_bytes, uint64_t rmaxinput, uint64_t rvard, uint64_t rrl, double ravgreadlength,
	dacc_overlap * out, uint64_t * nout, uint64_t * lmaxinput_out)
{
#if defined(DACC_REF_MAINSEL_A_EXCERPT) && defined(DACC_REF_MAINSEL_B_EXCERPT)
	*nout = 0;
	if ( lmaxinput_out ) *lmaxinput_out = 0;
	if ( !rmaxinput ) return 0;
	std::string raw;
	for ( uint64_t i = 0; i < n; ++i )
	{
		int32_t h[10] = { in[i].tlen, in[i].diffs, in[i].abpos, in[i].bbpos, in[i].aepos, in[i].bepos, static_cast<int32_t>(in[i].flags), in[i].aread, in[i].bread,
			static_cast<int32_t>(i) /* the padding word carries the record's index through the reference's copies */ };
		raw.append(reinterpret_cast<char const *>(h),40);
		for ( int64_t j = 0; j < static_cast<int64_t>(in[i].tlen)*trace_bytes; ++j ) raw.push_back(static_cast<char>((i*131+j*7)&0xFF));
	}
	std::string const lasfn = "ref_pile_select.las";
	libmaus2::aio::selRegistry()[lasfn] = raw;
	uint64_t const numthreads = 1;
	uint64_t const maxinput = rmaxinput;
	int64_t const tspace = trace_bytes == 1 ? 100 : 1000;
	try
	{
		#include DACC_REF_MAINSEL_A_EXCERPT
		// the names the loop body finds around it (:1309-1316 index of byte offsets per A read, :2111-2117 z and tid, RL, vard, avgreadlength)
		struct ByteIndex { uint64_t o[2]; uint64_t operator[](int64_t const i) const { return o[i]; } };
		libmaus2::autoarray::AutoArray< std::unique_ptr<ByteIndex> > Adalindex(1);
		Adalindex[0].reset(new ByteIndex); Adalindex[0]->o[0] = 0; Adalindex[0]->o[1] = raw.size();
		int64_t const minaread = 0, z = 0;
		uint64_t const tid = 0;
		std::vector<uint64_t> RL(1,rrl);
		uint64_t const vard = rvard;
		double const avgreadlength = ravgreadlength;
		#include DACC_REF_MAINSEL_B_EXCERPT
		for ( uint64_t i = 0; i < o_copypointers; ++i )
		{
			uint64_t const idx = static_cast<uint32_t>(libmaus2::dazzler::align::selGet32(copypointers[i].p+36));
			// the copy is the record, trace included
			uint64_t const s = 40 + static_cast<uint64_t>(in[idx].tlen)*trace_bytes;
			for ( uint64_t j = 40; j < s; ++j ) if ( copypointers[i].p[j] != static_cast<uint8_t>((idx*131+(j-40)*7)&0xFF) ) return -2;
			out[i] = in[idx];
		}
		*nout = o_copypointers;
		if ( lmaxinput_out ) *lmaxinput_out = lmaxinput;
	}
	catch(std::exception const & ex)
	{
		return -1;
	}
	libmaus2::aio::selRegistry().erase(lasfn);
	return 0;
#else
	(void)in; (void)n; (void)trace_bytes; (void)rmaxinput; (void)rvard; (void)rrl; (void)ravgreadlength; (void)out; (void)nout; (void)lmaxinput_out;
	return -9;
#endif
}


// The A reads of a run, src/daccord.cpp:1119-1224 (the -J part,parts and -I first,last branches with their parsing) and :1227
// (toparead), compiled from the reference's lines: arg is the two options' text, minaread / maxaread enter as the first and last
// A read of the overlap file (:1115-1116).  Returns 1 with the reference's message when its code throws.
int ref_read_interval(int64_t lasmin, int64_t lasmax, char const * J, char const * I, int64_t * minout, int64_t * topout, char * err, uint64_t errcap)
{
#if defined(DACC_REF_IVL_A_EXCERPT) && defined(DACC_REF_IVL_B_EXCERPT)
	struct Arg
	{
		char const * J; char const * I;
		bool uniqueArgPresent(std::string const & k) const { return k == "J" ? (J != 0) : (k == "I" ? (I != 0) : false); }
		std::string operator[](std::string const & k) const { return std::string(k == "J" ? J : I); }
	} arg; arg.J = J; arg.I = I;
	int64_t minaread = lasmin;
	int64_t maxaread = lasmax;
	try
	{
		#include DACC_REF_IVL_A_EXCERPT
	}
	catch(std::exception const & ex)
	{
		if ( err && errcap ) std::snprintf(err,errcap,"%s",ex.what());
		return 1;
	}
	#include DACC_REF_IVL_B_EXCERPT
	*minout = minaread; *topout = toparead;
	return 0;
#else
	(void)lasmin; (void)lasmax; (void)J; (void)I; (void)minout; (void)topout; (void)err; (void)errcap;
	return -9;
#endif
}


// The option defaults of src/daccord.cpp:106-169 (compiled from its lines): V, k, D, vard, d, w, a, f, m, e, l, minfilterfreq, maxfilterfreq
int ref_defaults(uint64_t * out)
{
#if defined(DACC_REF_DEFAULTS_EXCERPT)
	using namespace refdefaults;
	out[0] = getDefaultVerbose(); out[1] = getDefaultK(); out[2] = getDefaultMaxInput(); out[3] = getDefaultVarD(); out[4] = getDefaultMaxAlign();
	out[5] = getDefaultWindowSize(); out[6] = getDefaultAdvanceSize(); out[7] = getDefaultProduceFull(); out[8] = getDefaultMinWindowCoverage();
	out[9] = getDefaultMinWindowError(); out[10] = getDefaultMinLen(); out[11] = getDefaultMinFilterFreq(); out[12] = getDefaultMaxFilterFreq();
	return 0;
#else
	(void)out; return -9;
#endif
}

}
