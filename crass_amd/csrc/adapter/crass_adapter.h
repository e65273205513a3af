// crass_adapter.h — C++ host adapter: the reference's own seam, re-implemented over the C ABI.
//
// WorkHorse::parseSeqFiles (src/crass/WorkHorse.cpp:321-414) calls three functions; this header
// offers the same three shapes (names, argument meaning, error behaviour) so that swapping the
// include + link line is the whole integration:
//
//   int   searchFile(const char*, const options&, ReadMap*, StringCheck*, lookupTable& patternsHash,
//                    lookupTable& readsFound, time_t&)                       libcrispr.h:74-80
//   Vecstr* createNonRedundantSet(ReadMap&, StringCheck&, DR_Cluster_Map&, std::map<int,bool>&,
//                    GroupKmerMap&, int& nextFreeGID, const options&)        WorkHorse.h:111-112 (was a
//                    private member; state passed explicitly)
//   void  findSingletons(const char*, const options&, std::vector<std::string>*, lookupTable& readsFound,
//                    ReadMap*, StringCheck*, time_t&)                        libcrispr.h:86-92
//   void  addReadHolder(ReadMap*, StringCheck*, ReadHolder&)                 libcrispr.h:123-125
//
// The hand-off types below carry exactly the fields the downstream stages read
// (ReadHolder.h:440-451, StringCheck.h:52-71, Types.h:51-67); all search decisions come from the
// HIP engine (libcrass_hip.so) — nothing here re-runs the search on the CPU.
#pragma once
#include <ctime>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/crass_hip.h"

namespace crass_hip {

// crispr::exception (Exception.h:29-61): file, line, function, message
class exception : public std::runtime_error {
public:
    exception(const char *file, int line, const char *function, const std::string &message)
        : std::runtime_error(std::string("[ERROR]: ") + message + "\n" + file + " : " + std::to_string(line) + " : " + function) {}
};

// the hot-path subset of `options` (crassDefines.h:140-170) with the reference's field names
struct options {
    int logLevel = 1;
    unsigned int lowDRsize = 23, highDRsize = 47, lowSpacerSize = 26, highSpacerSize = 50;
    std::string output_fastq = "./";
    int kmer_clust_size = 6;
    unsigned int searchWindowLength = 8, minNumRepeats = 2;
    bool logToScreen = false;
    int covCutoff = 3;                 // -f, CRASS_DEF_COVCUTOFF (crassDefines.h:126)
    int cNodeKmerLength = 7;           // -K, CRASS_DEF_NODE_KMER_SIZE (:111)
    bool showSingles = false;          // -G (:138)
    bool longDescription = false;      // -L (:137)
};

typedef int StringToken;
typedef std::vector<unsigned int> StartStopList;
typedef std::vector<std::string> Vecstr;
typedef std::map<std::string, bool> lookupTable;

// StringCheck (StringCheck.h:52-71, StringCheck.cpp:46-81)
class StringCheck {
public:
    StringCheck() : mNextFreeToken(1) {}
    StringToken addString(const std::string &s) { mNextFreeToken++; mT2S_map[mNextFreeToken] = s; mS2T_map[s] = mNextFreeToken; return mNextFreeToken; }
    std::string getString(StringToken t) const;
    StringToken getToken(const std::string &s) const { auto it = mS2T_map.find(s); return it == mS2T_map.end() ? 0 : it->second; }
    StringToken mNextFreeToken;
    std::map<StringToken, std::string> mT2S_map;
    std::map<std::string, StringToken> mS2T_map;
};

// ReadHolder (ReadHolder.h:58-452): the per-read hand-off record, field for field
class ReadHolder {
public:
    ReadHolder() {}
    std::string getSeq() const { return RH_Seq; }
    std::string getHeader() const { return RH_Header; }
    std::string getComment() const { return RH_Comment; }
    std::string getQual() const { return RH_Qual; }
    bool getIsFasta() const { return RH_IsFasta; }
    bool getLowLexi() const { return RH_WasLowLexi; }
    StartStopList getStartStopList() const { return RH_StartStops; }
    unsigned int numRepeats() const { return (unsigned int)(RH_StartStops.size() / 2); }
    unsigned int numSpacers() const { return numRepeats() - 1; }
    unsigned int getRepeatLength() const { return (unsigned int)RH_RepeatLength; }
    int getSeqLength() const { return (int)RH_Seq.length(); }
    unsigned int front() const { return RH_StartStops.front(); }
    unsigned int back() const { return RH_StartStops.back(); }
    // ReadHolder::repeatStringAt (ReadHolder.cpp:77-100)
    std::string repeatStringAt(unsigned int i) const { return RH_Seq.substr(RH_StartStops[i], RH_StartStops[i + 1] - RH_StartStops[i] + 1); }

    std::string RH_Header, RH_Comment, RH_Qual, RH_Seq;
    bool RH_IsFasta = true;
    bool RH_WasLowLexi = false;
    StartStopList RH_StartStops;
    int RH_RepeatLength = 0;
};

typedef std::vector<ReadHolder *> ReadList;
typedef std::map<StringToken, ReadList *> ReadMap;                       // Types.h:51-56
typedef std::vector<StringToken> DR_Cluster;
typedef std::map<int, DR_Cluster *> DR_Cluster_Map;                       // Types.h:61-64
typedef std::map<int, std::map<std::string, int> *> GroupKmerMap;         // Types.h:65-67

// ---- the seam ----
int searchFile(const char *inputFastq, const options &opts, ReadMap *mReads, StringCheck *mStringCheck,
               lookupTable &patternsHash, lookupTable &readsFound, time_t &time_start);

Vecstr *createNonRedundantSet(ReadMap &mReads, StringCheck &mStringCheck, DR_Cluster_Map &mDR2GIDMap,
                              std::map<int, bool> &mGroupMap, GroupKmerMap &groupKmerCountsMap, int &nextFreeGID,
                              const options &opts);

void findSingletons(const char *inputFastq, const options &opts, std::vector<std::string> *nonRedundantPatterns,
                    lookupTable &readsFound, ReadMap *mReads, StringCheck *mStringCheck, time_t &startTime);

void addReadHolder(ReadMap *mReads, StringCheck *mStringCheck, ReadHolder &tmpReadholder);

// The stage right behind the search: int WorkHorse::findConsensusDRs(GroupKmerMap&, int& nextFreeGID)
// (WorkHorse.cpp:578-611, called at :403; a private member there, so the state it works on is passed explicitly).  Same
// effects on the hand-off state: mTrueDRs[GID] = laurenized true DR for every group that survives; groups that are
// killed / split / merged are removed from (set to NULL in) mDR2GIDMap and new groups appear under new GIDs; slaves that
// align in reverse get a new token (mStringCheck.addString) and their ReadList moves to it; read lists of slaves that could
// not be placed are cleared; every surviving ReadHolder has its start/stops repaired (ReadHolder::updateStartStops, partial
// repeats at the read ends included) and is reverse-complemented when the true DR is not in its laurenized form.
// Returns 0, or 1 where the reference's caller would have seen an exception (parseSeqFiles then returns 1).
int findConsensusDRs(ReadMap &mReads, StringCheck &mStringCheck, DR_Cluster_Map &mDR2GIDMap, std::map<int, std::string> &mTrueDRs,
                     GroupKmerMap &groupKmerCountsMap, int &nextFreeGID, int mMaxReadLength, const options &opts);

// Everything behind findConsensusDRs: WorkHorse::buildGraph, cleanGraph, makeSpacerGraphs,
// cleanSpacerGraphs, splitIntoContigs, generateFlankers, removeLowConfidenceNodeManagers and outputResults
// (WorkHorse.cpp:196-316, 454-577, 1642-1729, 1900-2249) over the hand-off state: writes <outdir>crass.crispr,
// Group_<gid>_<DR>.fa, Spacers_<gid>_<DR>_spacers.gv and crass.<timestamp>.keys.gv (crass_build_outputs, include/crass_hip.h)
// and prints what crass prints in these stages.  timeStamp / commandLine: WorkHorse's mTimeStamp / mCommandLine
// (crass.cpp:476-479,508-512).  Returns the number of CRISPRs written, or -1 on error.
int buildGraphsAndOutput(ReadMap &mReads, DR_Cluster_Map &mDR2GIDMap, std::map<int, std::string> &mTrueDRs, const options &opts,
                         const std::string &timeStamp, const std::string &commandLine);

// reads of every file searched so far stay resident on the GPU between searchFile() and
// findSingletons(); call this once the pipeline is past the search stage.
void releaseDeviceReads();
// which GPU the adapter uses (default 0, or $CRASS_HIP_DEVICE)
void setDevice(int device);
// a process that ends right behind the output stage (the command line) tells the adapter so: what the search stage leaves behind
// (the index of the inputs, the device contexts) is then left to the process's end instead of being freed on a thread beside the
// output stage, whose allocations wait for every unmapping (0.25 s of a 2.6 s run at 50 M reads).  Default: freed.
void leaveTeardownToProcessEnd(bool yes);
// several GPUs: every read set is sharded over them by contiguous read ranges (crass_hip_group_*, include/crass_hip.h); with
// searchAndRecruit the DR merge runs on the devices and the candidate DR strings cross in ONE RCCL all-gather issued by the
// engine.  local_copies: tests only — several contexts on one GPU, device copies instead of the collective.
void setDevices(const std::vector<int> &devices, bool local_copies = false);

// The first half of WorkHorse::parseSeqFiles (WorkHorse.cpp:336-398: searchFile over every file, createNonRedundantSet,
// findSingletons over every file) as ONE call that stays on the device(s) from the packed reads to the ordered hand-off —
// the form that lets the engine run the DR merge on the GPU and, with setDevices(), shard the reads over several GPUs.
// Effects on mReads / mStringCheck / mDR2GIDMap / mGroupMap / groupKmerCountsMap / nextFreeGID / patternsHash / readsFound
// are those of the three calls (same tokens, same GIDs, same vector orders); *nonRedundantPatterns receives the pattern
// list (may be NULL).  Returns the maximum read length (searchFile's return value, max over the files).  Needs a fresh
// StringCheck and ReadMap, which is what parseSeqFiles starts from.
int searchAndRecruit(const Vecstr &seqFiles, const options &opts, ReadMap *mReads, StringCheck *mStringCheck,
                     DR_Cluster_Map &mDR2GIDMap, std::map<int, bool> &mGroupMap, GroupKmerMap &groupKmerCountsMap,
                     int &nextFreeGID, lookupTable &patternsHash, lookupTable &readsFound, Vecstr *nonRedundantPatterns,
                     time_t &time_start);

// clean-up helpers with the ownership rules of WorkHorse::clearReadMap (WorkHorse.cpp:127-162)
void clearReadMap(ReadMap *m);

} // namespace crass_hip
