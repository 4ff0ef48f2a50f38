// pvlm_host_io.cpp — part of the C++ host mirror (pvlm_host.hpp): pose files and LiDAR clouds: util/FileIO.cpp:11-79, :168-191; sensors/Velodyne.cpp:92-172 (+ the part of pcl::io::loadPCDFile a PointXYZI cloud needs).
// Host logic only; every residual, Jacobian, distance and vote is produced by libpvlm.so on the GPU.
#include "pvlm_host_internal.hpp"

namespace pvlm {

// ================================================================================================
// pose files — util/FileIO.cpp:11-79 (ReadPoseT), :168-191 (ExportPoseT)
// ================================================================================================
bool ReadPoseT(std::string file_path, bool with_invalid, std::vector<Matrix3d>& rotation_list, std::vector<Vector3d>& trans_list,
               std::vector<std::string>& name_list) {
  std::ifstream in(file_path);
  if (!in.is_open()) { fprintf(stderr, "Fail to open %s\n", file_path.c_str()); return false; }
  while (!in.eof()) {
    Matrix3d R = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    Vector3d t = {INFINITY, INFINITY, INFINITY};
    std::string str;
    std::getline(in, str);
    std::vector<std::string> sub;
    { std::stringstream ss(str); std::string tmp; while (std::getline(ss, tmp, ' ')) sub.push_back(tmp); }  // SplitString(str, ' ')
    std::string curr_name;
    bool pose_valid = true;
    if (sub.size() == 13) { curr_name = sub[0]; sub.erase(sub.begin()); }
    if (sub.size() == 12) {
      for (const std::string& s : sub)
        if (s.find("inf") != std::string::npos || s.find("nan") != std::string::npos) { pose_valid = false; break; }
      if (pose_valid) {
        double v[12];
        for (int k = 0; k < 12; ++k) { std::stringstream ss(sub[k]); ss >> v[k]; }  // str2num<double>
        R = {v[0], v[1], v[2], v[4], v[5], v[6], v[8], v[9], v[10]};
        t = {v[3], v[7], v[11]};
      }
    }
    if (pose_valid || (!pose_valid && with_invalid)) { rotation_list.push_back(R); trans_list.push_back(t); name_list.push_back(curr_name); }
    if (in.peek() == EOF) break;
  }
  return true;
}

void ExportPoseT(const std::string file_path, const std::vector<Matrix3d>& rotation_list, const std::vector<Vector3d>& trans_list,
                 const std::vector<std::string>& name_list, int precision) {
  std::ofstream out(file_path);
  if (!out.is_open()) { fprintf(stderr, "Fail to write %s\n", file_path.c_str()); return; }
  out << std::setprecision(precision);
  for (size_t i = 0; i < rotation_list.size() && i < trans_list.size(); i++) {
    if (i < name_list.size()) out << name_list[i] << " ";
    const Matrix3d& R = rotation_list[i]; const Vector3d& t = trans_list[i];
    out << R[0] << " " << R[1] << " " << R[2] << " " << t[0] << " " << R[3] << " " << R[4] << " " << R[5] << " " << t[1] << " " << R[6] << " " << R[7]
        << " " << R[8] << " " << t[2] << std::endl;
  }
}

// ================================================================================================
// LoadLidar — sensors/Velodyne.cpp:92-172 (+ the part of pcl::io::loadPCDFile a PointXYZI cloud needs)
// ================================================================================================
namespace {
// LZF decompression (the codec of PCD "binary_compressed"): control byte < 32 = literal run of ctrl + 1 bytes; otherwise
// a back reference of length (ctrl >> 5) + 2 (length 7 reads one extension byte) at offset ((ctrl & 31) << 8 | next) + 1.
bool LzfDecompress(const unsigned char* in, size_t in_len, unsigned char* out, size_t out_len) {
  size_t ip = 0, op = 0;
  while (ip < in_len) {
    unsigned ctrl = in[ip++];
    if (ctrl < 32) {
      const size_t run = ctrl + 1;
      if (ip + run > in_len || op + run > out_len) return false;
      std::memcpy(out + op, in + ip, run); ip += run; op += run;
    } else {
      size_t len = ctrl >> 5;
      if (len == 7) { if (ip >= in_len) return false; len += in[ip++]; }
      if (ip >= in_len) return false;
      const size_t off = ((size_t)(ctrl & 0x1f) << 8) + in[ip++] + 1;
      len += 2;
      if (off > op || op + len > out_len) return false;
      for (size_t k = 0; k < len; ++k, ++op) out[op] = out[op - off];   // may overlap: byte by byte
    }
  }
  return op == out_len;
}

struct PcdField { std::string name; int size = 4; char type = 'F'; int count = 1; size_t offset = 0; };

bool ReadPcd(const std::string& path, PointCloud& cloud) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  std::vector<PcdField> fields;
  size_t points = 0, width = 0, height = 1;
  std::string data_mode, line;
  while (std::getline(f, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    if (line.empty() || line[0] == '#') continue;
    std::istringstream ls(line);
    std::string key; ls >> key;
    if (key == "FIELDS") { std::string n; while (ls >> n) { PcdField fd; fd.name = n; fields.push_back(fd); } }
    else if (key == "SIZE") { for (PcdField& fd : fields) ls >> fd.size; }
    else if (key == "TYPE") { for (PcdField& fd : fields) ls >> fd.type; }
    else if (key == "COUNT") { for (PcdField& fd : fields) ls >> fd.count; }
    else if (key == "WIDTH") ls >> width;
    else if (key == "HEIGHT") ls >> height;
    else if (key == "POINTS") ls >> points;
    else if (key == "DATA") { ls >> data_mode; break; }
  }
  if (fields.empty() || data_mode.empty()) return false;
  if (points == 0) points = width * height;
  size_t stride = 0;
  for (PcdField& fd : fields) { fd.offset = stride; stride += (size_t)fd.size * fd.count; }
  int ix = -1, iy = -1, iz = -1, ii = -1;
  for (size_t k = 0; k < fields.size(); ++k) {
    if (fields[k].name == "x") ix = (int)k; else if (fields[k].name == "y") iy = (int)k; else if (fields[k].name == "z") iz = (int)k;
    else if (fields[k].name == "intensity") ii = (int)k;
  }
  if (ix < 0 || iy < 0 || iz < 0) return false;
  for (int k : {ix, iy, iz}) if (fields[k].type != 'F' || fields[k].size != 4) return false;   // PointXYZI: float32 coordinates
  for (const PcdField& fd : fields) if (fd.size <= 0 || fd.size > 8 || fd.count <= 0 || fd.count > 4096) return false;
  // a header must not make us allocate more than the file can hold (corrupt / hostile POINTS, WIDTH x HEIGHT)
  const std::streampos body = f.tellg();
  f.seekg(0, std::ios::end);
  const size_t remaining = body < 0 ? 0 : (size_t)(f.tellg() - body);
  f.seekg(body);
  if (stride == 0) return false;
  if (data_mode == "ascii" && points > remaining) return false;                         // >= 1 byte per point
  else if (data_mode == "binary" && points > remaining / stride) return false;
  else if (data_mode == "binary_compressed") {
    if (remaining < 8 || points > 0xffffffffull / stride) return false;                 // the block length is a uint32
    if (points * stride / 100 > remaining) return false;                                // LZF expands at most 264 bytes per 3
  } else if (data_mode != "ascii" && data_mode != "binary") return false;
  cloud.assign(points, PointXYZI{0, 0, 0, 0});
  auto as_float = [](const PcdField& fd, const unsigned char* p) -> float {
    if (fd.type == 'F' && fd.size == 4) { float v; std::memcpy(&v, p, 4); return v; }
    if (fd.type == 'F' && fd.size == 8) { double v; std::memcpy(&v, p, 8); return (float)v; }
    if (fd.type == 'U' && fd.size == 1) return (float)*p;
    if (fd.type == 'U' && fd.size == 2) { uint16_t v; std::memcpy(&v, p, 2); return (float)v; }
    if (fd.type == 'U' && fd.size == 4) { uint32_t v; std::memcpy(&v, p, 4); return (float)v; }
    if (fd.type == 'I' && fd.size == 4) { int32_t v; std::memcpy(&v, p, 4); return (float)v; }
    return 0.f;
  };
  if (data_mode == "ascii") {
    for (size_t i = 0; i < points; ++i) {
      if (!std::getline(f, line)) return false;
      std::istringstream ls(line);
      for (size_t k = 0; k < fields.size(); ++k)
        for (int c = 0; c < fields[k].count; ++c) {
          std::string tok; ls >> tok;
          if (c > 0) continue;
          float v = (tok == "nan" || tok == "-nan" || tok == "NaN") ? NAN : (float)std::strtod(tok.c_str(), nullptr);
          if ((int)k == ix) cloud[i].x = v; else if ((int)k == iy) cloud[i].y = v; else if ((int)k == iz) cloud[i].z = v; else if ((int)k == ii) cloud[i].intensity = v;
        }
    }
    return true;
  }
  std::vector<unsigned char> raw;
  if (data_mode == "binary") {
    raw.resize(points * stride);
    f.read(reinterpret_cast<char*>(raw.data()), (std::streamsize)raw.size());
    if ((size_t)f.gcount() != raw.size()) return false;
    for (size_t i = 0; i < points; ++i) {
      const unsigned char* p = raw.data() + i * stride;
      cloud[i].x = as_float(fields[ix], p + fields[ix].offset); cloud[i].y = as_float(fields[iy], p + fields[iy].offset);
      cloud[i].z = as_float(fields[iz], p + fields[iz].offset);
      if (ii >= 0) cloud[i].intensity = as_float(fields[ii], p + fields[ii].offset);
    }
    return true;
  }
  if (data_mode == "binary_compressed") {
    uint32_t csize = 0, usize = 0;
    f.read(reinterpret_cast<char*>(&csize), 4); f.read(reinterpret_cast<char*>(&usize), 4);
    if (!f || usize != points * stride || remaining < 8 || csize > remaining - 8) return false;
    std::vector<unsigned char> comp(csize);
    f.read(reinterpret_cast<char*>(comp.data()), csize);
    if ((size_t)f.gcount() != csize) return false;
    raw.resize(usize);
    if (!LzfDecompress(comp.data(), csize, raw.data(), usize)) return false;
    // the uncompressed block is field-major (all x, all y, ...)
    size_t base = 0;
    std::vector<size_t> fbase(fields.size());
    for (size_t k = 0; k < fields.size(); ++k) { fbase[k] = base; base += (size_t)fields[k].size * fields[k].count * points; }
    for (size_t i = 0; i < points; ++i) {
      auto at = [&](int k) { return raw.data() + fbase[k] + i * (size_t)fields[k].size * fields[k].count; };
      cloud[i].x = as_float(fields[ix], at(ix)); cloud[i].y = as_float(fields[iy], at(iy)); cloud[i].z = as_float(fields[iz], at(iz));
      if (ii >= 0) cloud[i].intensity = as_float(fields[ii], at(ii));
    }
    return true;
  }
  return false;
}
}  // namespace

bool Velodyne::LoadLidar(std::string file_path) {
  if (file_path.empty()) file_path = name;
  const std::string::size_type pos = file_path.rfind('.');
  const std::string type = pos == std::string::npos ? "" : file_path.substr(pos);
  if (type != ".pcd") { fprintf(stderr, "unknown point cloud format, only .pcd is mirrored (the reference also reads .ply)\n"); return false; }
  PointCloud raw;
  if (!ReadPcd(file_path, raw)) { fprintf(stderr, "Fail to load lidar data at %s\n", file_path.c_str()); return false; }
  name = file_path;
  cloud.clear();
  for (const PointXYZI& p : raw) {
    if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;      // pcl::removeNaNFromPointCloud
    const float dis = p.x * p.x + p.y * p.y + p.z * p.z;                                  // removeClosedPointCloud(0.5), float
    if (dis < 0.5f * 0.5f) continue;
    // T_cam_lidar (:127-132): X right, Y forward, Z up  ->  X right, Y down, Z forward
    cloud.push_back({p.x, -p.z, p.y, p.intensity});
  }
  if (cloud.size() < 4000) { fprintf(stderr, "lidar %d is invalid, only %zu points in point cloud\n", id, cloud.size()); valid = false; }
  return true;
}


}  // namespace pvlm
