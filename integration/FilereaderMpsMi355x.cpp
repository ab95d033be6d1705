// FilereaderMpsMi355x.cpp — drop-in replacement TU for io/FilereaderMps.cpp (MPS ingest).
//
// Same class, same two member functions (io/FilereaderMps.h:17-25); a maintainer swaps this file for
// highs/io/FilereaderMps.cpp in the build and links libpdlp_mi355x.so.  readModelFromFile hands free-format files to
// the library's multi-threaded reader (pdlp_mi355x_read_mps, include/pdlp_mi355x.h) and fills HighsModel exactly as
// free_format_parser::HMpsFF::loadProblem does (io/HMpsFF.cpp:82-133).  Everything that reader does not take on stays
// the reference's own code, literally: the original TU is kept in the build under another class name
// (-DFilereaderMps=FilereaderMpsReference when compiling highs/io/FilereaderMps.cpp, see integration/build_dropin.sh)
// and is called for names with spaces / mps_parser_type_free = false (fixed-column reader, io/HMPSIO.cpp), for gzip
// streams on a system without libz (HMpsFF through zstr) and for writing (writeModelAsMps).
// Highs::readModel -> Filereader::getFilereader -> this TU -> Highs::passModel is otherwise untouched.
#include "io/FilereaderMps.h"

#include "lp_data/HighsLp.h"
#include "lp_data/HighsLpUtils.h"
#include "lp_data/HighsModelUtils.h"
#include "pdlp_mi355x.h"

// the reference's io/FilereaderMps.cpp, compiled with -DFilereaderMps=FilereaderMpsReference
class FilereaderMpsReference : public Filereader {
 public:
  FilereaderRetcode readModelFromFile(const HighsOptions& options, const std::string filename, HighsModel& model);
  HighsStatus writeModelToFile(const HighsOptions& options, const std::string filename, const HighsModel& model);
};

namespace {

// one line of the reader's warning text per highsLogUser call, as the reference logs its own
void logWarnings(const HighsLogOptions& log_options, const char* text) {
  if (!text) return;
  std::string line;
  for (const char* p = text;; ++p) {
    if (*p == '\n' || *p == '\0') {
      if (!line.empty()) highsLogUser(log_options, HighsLogType::kWarning, "%s\n", line.c_str());
      line.clear();
      if (*p == '\0') break;
    } else {
      line.push_back(*p);
    }
  }
}

void namesFromPool(const char* pool, const int64_t* start, HighsInt count, std::vector<std::string>& names) {
  names.clear();
  if (!pool || !start) return;
  names.reserve(count);
  for (HighsInt i = 0; i < count; i++) names.emplace_back(pool + start[i]);
}

// HMpsFF::loadProblem, io/HMpsFF.cpp:82-133
void fillModel(const pdlp_mps_model_t& m, HighsModel& model) {
  HighsLp& lp = model.lp_;
  HighsHessian& hessian = model.hessian_;
  const pdlp_problem_t& P = m.lp;
  lp.num_col_ = P.num_col;
  lp.num_row_ = P.num_row;
  lp.sense_ = P.sense < 0 ? ObjSense::kMaximize : ObjSense::kMinimize;
  lp.offset_ = P.offset;
  lp.a_matrix_.format_ = MatrixFormat::kColwise;
  lp.a_matrix_.start_.assign(P.a_start, P.a_start + P.num_col + 1);
  lp.a_matrix_.index_.assign(P.a_index, P.a_index + P.num_nz);
  lp.a_matrix_.value_.assign(P.a_value, P.a_value + P.num_nz);
  lp.col_cost_.assign(P.col_cost, P.col_cost + P.num_col);
  lp.col_lower_.assign(P.col_lower, P.col_lower + P.num_col);
  lp.col_upper_.assign(P.col_upper, P.col_upper + P.num_col);
  lp.row_lower_.assign(P.row_lower, P.row_lower + P.num_row);
  lp.row_upper_.assign(P.row_upper, P.row_upper + P.num_row);
  lp.objective_name_ = m.objective_name ? m.objective_name : "";
  namesFromPool(m.row_name_pool, m.row_name_start, P.num_row, lp.row_names_);
  namesFromPool(m.col_name_pool, m.col_name_start, P.num_col, lp.col_names_);
  lp.integrality_.clear();
  if (m.num_integrality > 0) {
    lp.integrality_.resize(P.num_col);
    for (HighsInt j = 0; j < P.num_col; j++) lp.integrality_[j] = static_cast<HighsVarType>(m.integrality[j]);
  }
  if (m.hessian_dim > 0) {
    hessian.dim_ = m.hessian_dim;
    hessian.format_ = HessianFormat::kSquare;
    const HighsInt q_nz = m.hessian_start[m.hessian_dim];
    hessian.start_.assign(m.hessian_start, m.hessian_start + m.hessian_dim + 1);
    hessian.index_.assign(m.hessian_index, m.hessian_index + q_nz);
    hessian.value_.assign(m.hessian_value, m.hessian_value + q_nz);
  } else {
    hessian.clear();
  }
  lp.objective_name_ = findModelObjectiveName(&lp, &hessian);
  lp.cost_row_location_ = m.cost_row_location;
}

}  // namespace

FilereaderRetcode FilereaderMps::readModelFromFile(const HighsOptions& options, const std::string filename,
                                                   HighsModel& model) {
  if (options.mps_parser_type_free) {
    pdlp_mps_model_t m;
    const double limit = options.time_limit < kHighsInf && options.time_limit > 0 ? options.time_limit : 0.0;  // io/FilereaderMps.cpp:30-31
    const int rc = pdlp_mi355x_read_mps_timed(filename.c_str(), options.threads, limit, &m);  // threads = 0: automatic
    if (rc == 0) {
      logWarnings(options.log_options, m.warnings);
      fillModel(m, model);
      const bool warning = m.warning_issued != 0;
      pdlp_mi355x_free_mps_model(&m);
      model.lp_.ensureColwise();
      return warning ? FilereaderRetcode::kWarning : FilereaderRetcode::kOk;
    }
    if (rc == 2) return FilereaderRetcode::kFileNotFound;
    if (rc == 1) {
      highsLogUser(options.log_options, HighsLogType::kError, "%s\n", pdlp_mi355x_last_error());
      return FilereaderRetcode::kParserError;
    }
    if (rc == 5) {
      highsLogUser(options.log_options, HighsLogType::kWarning,
                   "Free format reader reached time_limit while parsing the input file\n");
      return FilereaderRetcode::kTimeout;
    }
    if (rc == 3) {
      // names with spaces: the reference's free-format reader would find the same and hand over to the fixed-column
      // reader (io/FilereaderMps.cpp:45-49) — go there directly instead of parsing the file a second time
      highsLogUser(options.log_options, HighsLogType::kWarning,
                   "Free format reader has detected row/col names with spaces: switching to fixed format parser\n");
      HighsOptions fixed = options;
      fixed.mps_parser_type_free = false;
      const FilereaderRetcode rcFixed = FilereaderMpsReference().readModelFromFile(fixed, filename, model);
      // (the reference marks the read as one that issued a warning in this case)
      return rcFixed == FilereaderRetcode::kOk ? FilereaderRetcode::kWarning : rcFixed;
    }
    // 4: a gzip stream and no libz for the library -> the reference's TU
  }
  return FilereaderMpsReference().readModelFromFile(options, filename, model);
}

HighsStatus FilereaderMps::writeModelToFile(const HighsOptions& options, const std::string filename,
                                            const HighsModel& model) {
  return FilereaderMpsReference().writeModelToFile(options, filename, model);
}
