// pdlp_mps.hpp — multi-threaded MPS ingest: the data format on the caller's side of the path.
//
// Builds the model HiGHS' free-format MPS parser builds (io/HMpsFF.cpp, reached from
// io/FilereaderMps.cpp:24-58 <- Highs::readModel): same section rules, same treatment of duplicate / undefined
// names, zero coefficients, RANGES signs, default bounds of integer columns, objective offset = -RHS of the
// cost row, matrix entries in FILE ORDER inside each column.  Not a port: the reference reads the file line by
// line through an istream with std::string words and an unordered_map per lookup; here the file is mapped, cut at
// line boundaries into one piece per host thread, every section's lines are tokenised and looked up in parallel
// (read-only sharded hash tables over views into the mapping), and only the order-dependent rules (first
// occurrence wins, duplicate bounds, new columns met in BOUNDS) run in one short sequential pass over compact
// records.  Host-only C++ — no device code.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace pdlp {
namespace mps {

enum ReadStatus : int32_t {
  kReadOk = 0,
  kReadError = 1,         // malformed file (message in Model::error)
  kReadNotFound = 2,      // cannot open / map the file
  kReadFixedFormat = 3,   // names with spaces: a fixed-column reader is needed (FreeFormatParserReturnCode::kFixedFormat)
  kReadCompressed = 4,    // gzip stream and no zlib on this system (the reader inflates gzip files itself when it finds libz)
  kReadTimeout = 5,       // the time limit passed between two phases of the read (FreeFormatParserReturnCode::kTimeout)
};

// HighsVarType (lp_data/HConst.h)
enum VarType : uint8_t { kContinuous = 0, kInteger = 1, kSemiContinuous = 2, kSemiInteger = 3 };

struct Model {
  int32_t numCol = 0, numRow = 0;
  int32_t sense = 1;            // +1 min, -1 max
  double offset = 0.0;
  int32_t costRowLocation = -1; // lp.cost_row_location_
  std::vector<int32_t> aStart, aIndex;
  std::vector<double> aValue, colCost, colLower, colUpper, rowLower, rowUpper;
  std::vector<uint8_t> integrality;  // empty when every column is continuous
  // Hessian as the parser leaves it: square, column-wise, entries in file order (fillHessian, HMpsFF.cpp:177-216)
  int32_t qDim = 0;
  std::vector<int32_t> qStart, qIndex;
  std::vector<double> qValue;
  std::string modelName, objectiveName;
  // names as one pool of NUL-terminated strings + start offsets; EMPTY when the file has duplicate names
  // (HMpsFF.cpp:63-80 clears the name arrays then)
  std::string colNamePool, rowNamePool;
  std::vector<int64_t> colNameStart, rowNameStart;
  int32_t numWarnings = 0;      // warning classes met (all of them)
  bool warningIssued = false;   // HMpsFF::warning_issued_ at the end of the read: the reference's return status
  std::string warnings;  // one line per warning class, as the reference logs them
  std::string error;
  int32_t threads = 0;
  int64_t fileBytes = 0;
  double seconds = 0.0;
};

// numThreads <= 0: one per hardware thread (at least 1 MB of file each, at most 64); > 0: exactly that many pieces.
// timeLimit <= 0 or infinite: none (HMpsFF::timeout(), io/HMpsFF.cpp:218-220: checked between the phases here, line by line there)
ReadStatus readMps(const std::string& path, int numThreads, Model& out, double timeLimit = 0.0);

// Lower triangle (column-wise, rows ascending, duplicates summed, (q_ij + q_ji)/2 for a square input) of the
// parser's Hessian — what normaliseHessian (model/HighsHessianUtils.cpp:320) leaves and what pdlp_problem_t wants.
void lowerTriangle(const Model& m, std::vector<int32_t>& start, std::vector<int32_t>& index, std::vector<double>& value);

}  // namespace mps
}  // namespace pdlp
