// libfluid_amd/host/types.h -- the value types the host classes exchange with their callers: vec2 / vec3, grid3, mac_grid, source,
// mesh. The callers of lukedan/libfluid hand `fluid::simulation` their OWN `fluid::vec3d / vec3s`, `fluid::grid3`,
// `fluid::mac_grid::cell`, `fluid::source` and `fluid::mesh` objects (testbed/main.cpp:50-125,139,167-176,203-232;
// plugins/maya/nodes/grid_node.cpp:256-274,295-303,330-366; voxelizer_node.cpp:222-343), so a drop-in has to take exactly those:
//
//   * When the reference's headers are on the include path (`-I <libfluid>/include`, which every host of the reference has),
//     `fluid_amd::vec3d`, `grid3`, `mac_grid`, `source`, `mesh` ARE the reference's types (aliases, nothing converted or copied)
//     and `fluid_amd::vec_ops` is `fluid::vec_ops`. Only headers are needed: nothing here calls into the reference's .cpp files.
//   * Without them (a host that has no libfluid checkout; the GPU box of this repository's tests) the same names are
//     self-contained types with the same members, defined below.
//
// `LFA_HOST_OWN_TYPES` forces the second form. The host classes are written against the common subset of the two.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <optional>
#include <type_traits>
#include <utility>
#include <vector>

#if !defined(LFA_HOST_OWN_TYPES) && defined(__has_include)
	// (<fluid/misc.h> is the witness of a real checkout: the reference's vec.h includes it, shim_standalone/ has no such file)
#	if __has_include(<fluid/misc.h>) && __has_include(<fluid/math/vec.h>) && __has_include(<fluid/data_structures/grid.h>) && \
		__has_include(<fluid/mac_grid.h>) && __has_include(<fluid/data_structures/source.h>) && __has_include(<fluid/data_structures/mesh.h>)
#		define LFA_HOST_REFERENCE_TYPES 1
#	endif
#endif

#ifdef LFA_HOST_REFERENCE_TYPES

#include <fluid/math/vec.h>
#include <fluid/data_structures/grid.h>
#include <fluid/mac_grid.h>
#include <fluid/data_structures/source.h>
#include <fluid/data_structures/mesh.h>

namespace fluid_amd {
	namespace vec_ops = ::fluid::vec_ops;
	using ::fluid::vec2;
	using ::fluid::vec3;
	using ::fluid::vec2d;
	using ::fluid::vec2s;
	using ::fluid::vec3d;
	using ::fluid::vec3f;
	using ::fluid::vec3i;
	using ::fluid::vec3s;
	using ::fluid::grid3;
	using ::fluid::mac_grid;
	using ::fluid::source;
	using ::fluid::mesh;
}

#else  // ---------------------------------------------------------------------------------------- self-contained types

namespace fluid_amd {
	/// Small fixed vectors with the members of fluid::vec<2 / 3, T> (include/fluid/math/vec.h:188-511).
	template <typename T> struct vec2 {
		using value_type = T;
		constexpr static std::size_t dimensionality = 2;
		constexpr static std::size_t size() { return 2; }
		T x{}, y{};
		vec2() = default;
		vec2(T a, T b) : x(a), y(b) {}
		template <typename U> explicit vec2(const vec2<U> &o) : x(static_cast<T>(o.x)), y(static_cast<T>(o.y)) {}
		T &at(std::size_t i) { return (&x)[i]; }
		T at(std::size_t i) const { return (&x)[i]; }
		T &operator[](std::size_t i) { return (&x)[i]; }
		T operator[](std::size_t i) const { return (&x)[i]; }
	};
	template <typename T> struct vec3 {
		using value_type = T;
		constexpr static std::size_t dimensionality = 3;
		constexpr static std::size_t size() { return 3; }
		T x{}, y{}, z{};
		vec3() = default;
		template <typename A, typename B, typename C> vec3(A &&a, B &&b, C &&c) { x = std::forward<A>(a); y = std::forward<B>(b); z = std::forward<C>(c); }
		template <typename U> explicit vec3(const vec3<U> &o) : x(static_cast<T>(o.x)), y(static_cast<T>(o.y)), z(static_cast<T>(o.z)) {}
		template <std::size_t I> static vec3 axis() { vec3 r; r[I] = static_cast<T>(1); return r; }
		T &at(std::size_t i) { return (&x)[i]; }
		T at(std::size_t i) const { return (&x)[i]; }
		T &operator[](std::size_t i) { return (&x)[i]; }
		T operator[](std::size_t i) const { return (&x)[i]; }
		vec3 &operator+=(const vec3 &o) { x += o.x; y += o.y; z += o.z; return *this; }
		vec3 &operator-=(const vec3 &o) { x -= o.x; y -= o.y; z -= o.z; return *this; }
		template <typename U> vec3 &operator*=(const U &s) { x *= s; y *= s; z *= s; return *this; }
		template <typename U> vec3 &operator/=(const U &s) { x /= s; y /= s; z /= s; return *this; }
		friend vec3 operator+(const vec3 &a, const vec3 &b) { return vec3(a) += b; }
		friend vec3 operator-(const vec3 &a, const vec3 &b) { return vec3(a) -= b; }
		friend vec3 operator-(const vec3 &a) { return vec3(-a.x, -a.y, -a.z); }
		template <typename U, typename = std::enable_if_t<std::is_arithmetic_v<U>>> friend vec3 operator*(vec3 a, const U &s) { return a *= s; }
		template <typename U, typename = std::enable_if_t<std::is_arithmetic_v<U>>> friend vec3 operator*(const U &s, vec3 a) { return a *= s; }
		template <typename U, typename = std::enable_if_t<std::is_arithmetic_v<U>>> friend vec3 operator/(vec3 a, const U &s) { return a /= s; }
		template <typename D = T, typename = std::enable_if_t<std::is_integral_v<D>>>
		friend bool operator==(const vec3 &a, const vec3 &b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
		template <typename D = T, typename = std::enable_if_t<std::is_integral_v<D>>>
		friend bool operator!=(const vec3 &a, const vec3 &b) { return !(a == b); }
		T squared_length() const { T r{}; r += x * x; r += y * y; r += z * z; return r; }
		template <typename D = T, typename = std::enable_if_t<std::is_floating_point_v<D>>> T length() const { return std::sqrt(squared_length()); }
		template <typename D = T, typename = std::enable_if_t<std::is_floating_point_v<D>>>
		std::optional<vec3> normalized_checked(T eps = static_cast<T>(1e-6)) const {
			const T sq = squared_length();
			if (sq <= eps * eps) return std::nullopt;
			return *this / std::sqrt(sq);
		}
		template <typename D = T, typename = std::enable_if_t<std::is_floating_point_v<D>>>
		vec3 normalized_unchecked() const { return *this / length(); }
	};
	using vec2d = vec2<double>;
	using vec2s = vec2<std::size_t>;
	using vec3d = vec3<double>;
	using vec3f = vec3<float>;
	using vec3i = vec3<int>;
	using vec3s = vec3<std::size_t>;

	/// The entry points of fluid::vec_ops callers use (include/fluid/math/vec.h:17-187,540-547).
	namespace vec_ops {
		template <typename Vec> typename Vec::value_type dot(const Vec &a, const Vec &b) {
			typename Vec::value_type r{};
			for (std::size_t i = 0; i < Vec::size(); ++i) r += a[i] * b[i];
			return r;
		}
		template <typename T> vec3<T> cross(const vec3<T> &a, const vec3<T> &b) {
			return vec3<T>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
		}
		template <typename Func, typename First, typename... Others> void for_each(Func &&f, First &&first, Others &&...others) {
			for (std::size_t i = 0; i < std::decay_t<First>::size(); ++i) f(first[i], others[i]...);
		}
		template <typename Res, typename Func, typename... Vecs> void apply_to(Res &out, const Func &f, const Vecs &...v) {
			for (std::size_t i = 0; i < Res::size(); ++i) out[i] = f(v[i]...);
		}
		template <typename Res, typename Func, typename... Vecs> Res apply(const Func &f, const Vecs &...v) {
			Res r;
			apply_to(r, f, v...);
			return r;
		}
		namespace memberwise {
			template <typename Vec> Vec mul(const Vec &a, const Vec &b) { return apply<Vec>([](auto p, auto q) { return p * q; }, a, b); }
			template <typename Vec> Vec div(const Vec &a, const Vec &b) { return apply<Vec>([](auto p, auto q) { return p / q; }, a, b); }
		}
	}

	/// Dense x-fastest 3-D array with the members of fluid::grid3 (include/fluid/data_structures/grid.h:13-291; march_cells,
	/// which only the reference's own collision code uses, is not part of it).
	template <typename Cell> class grid3 {
	public:
		using size_type = vec3s;
		grid3() = default;
		explicit grid3(size_type size) : grid3(size, Cell{}) {}
		grid3(size_type size, const Cell &c) : _cells(get_array_size(size), c), _size(size) {}
		Cell &at(size_type i) { return _cells[index_to_raw(i)]; }
		const Cell &at(size_type i) const { return _cells[index_to_raw(i)]; }
		template <typename A, typename B, typename C> Cell &operator()(A &&x, B &&y, C &&z) { return at(size_type(x, y, z)); }
		template <typename A, typename B, typename C> const Cell &operator()(A &&x, B &&y, C &&z) const { return at(size_type(x, y, z)); }
		Cell &operator()(size_type i) { return at(i); }
		const Cell &operator()(size_type i) const { return at(i); }
		Cell &at_raw(std::size_t raw) { return _cells[raw]; }
		const Cell &at_raw(std::size_t raw) const { return _cells[raw]; }
		Cell &operator[](std::size_t raw) { return _cells[raw]; }
		const Cell &operator[](std::size_t raw) const { return _cells[raw]; }
		size_type get_size() const { return _size; }
		void fill(const Cell &c) { std::fill(_cells.begin(), _cells.end(), c); }
		bool is_border_cell(size_type i) const {
			return i.x == 0 || i.y == 0 || i.z == 0 || i.x == _size.x - 1 || i.y == _size.y - 1 || i.z == _size.z - 1;
		}
		template <typename Cb> void for_each(Cb &&cb) { for_each_in_range_unchecked(std::forward<Cb>(cb), size_type(), _size); }
		template <typename Cb> void for_each_in_range_unchecked(Cb &&cb, size_type lo, size_type hi) {
			for (std::size_t z = lo.z; z < hi.z; ++z)
				for (std::size_t y = lo.y; y < hi.y; ++y)
					for (std::size_t x = lo.x; x < hi.x; ++x) cb(size_type(x, y, z), at(size_type(x, y, z)));
		}
		template <typename Cb> void for_each_in_range_checked(Cb &&cb, size_type lo, size_type hi) {
			hi = size_type(std::min(hi.x, _size.x), std::min(hi.y, _size.y), std::min(hi.z, _size.z));
			for_each_in_range_unchecked(std::forward<Cb>(cb), lo, hi);
		}
		template <typename Cb> void for_each_in_range_checked(Cb &&cb, size_type center, size_type dmin, size_type dmax) {
			size_type lo(center.x < dmin.x ? 0 : center.x - dmin.x, center.y < dmin.y ? 0 : center.y - dmin.y,
			             center.z < dmin.z ? 0 : center.z - dmin.z);
			for_each_in_range_checked(std::forward<Cb>(cb), lo, center + dmax + size_type(1, 1, 1));
		}
		std::size_t index_to_raw(size_type i) const { return i.x + _size.x * (i.y + _size.y * i.z); }
		size_type index_from_raw(std::size_t r) const {
			size_type v;
			v.x = r % _size.x; r /= _size.x;
			v.y = r % _size.y; r /= _size.y;
			v.z = r % _size.z;
			return v;
		}
		static std::size_t get_array_size(size_type size) { return size.x * size.y * size.z; }
	private:
		std::vector<Cell> _cells;
		size_type _size;
	};

	/// fluid::mac_grid (include/fluid/mac_grid.h:12-73): 32-byte cells, out-of-range == solid. (get_face_samples belongs to the
	/// reference's CPU transfer code and has no host-side counterpart here: the transfers run on the device.)
	class mac_grid {
	public:
		struct cell {
			enum class type : unsigned char { air = 0x1, fluid = 0x2, solid = 0x4 };
			vec3d velocities_posface;
			type cell_type = type::air;
		};
		mac_grid() = default;
		explicit mac_grid(vec3s n) : _grid(n) {}
		cell *get_cell(vec3s i) {
			vec3s n = _grid.get_size();
			return (i.x >= n.x || i.y >= n.y || i.z >= n.z) ? nullptr : &_grid(i);
		}
		const cell *get_cell(vec3s i) const { return const_cast<mac_grid*>(this)->get_cell(i); }
		std::pair<cell*, cell::type> get_cell_and_type(vec3s i) {
			if (cell *c = get_cell(i)) return {c, c->cell_type};
			return {nullptr, cell::type::solid};
		}
		std::pair<const cell*, cell::type> get_cell_and_type(vec3s i) const {
			if (const cell *c = get_cell(i)) return {c, c->cell_type};
			return {nullptr, cell::type::solid};
		}
		grid3<cell> &grid() { return _grid; }
		const grid3<cell> &grid() const { return _grid; }
	protected:
		grid3<cell> _grid;
	};

	/// fluid::source (include/fluid/data_structures/source.h:12-22).
	class source {
	public:
		std::vector<vec3s> cells;
		vec3d velocity;
		std::size_t target_density_cubic_root = 2;
		bool active = true, coerce_velocity = false;
	};
}

#endif  // LFA_HOST_REFERENCE_TYPES

namespace fluid_amd {
	static_assert(sizeof(vec3d) == 24 && sizeof(vec3s) == 24 && sizeof(vec3i) == 12, "vec3 must be three packed coordinates");
	static_assert(sizeof(mac_grid::cell) == 32, "cell layout must match the reference (32-B AoS, include/fluid/mac_grid.h:15-27)");

	namespace detail {
		/// Number of cells / pointer to the x-fastest cell storage of a grid3 of either flavour (the reference's has no data()).
		template <typename Cell> std::size_t cell_count(const grid3<Cell> &g) { return grid3<Cell>::get_array_size(g.get_size()); }
		template <typename Cell> Cell *cell_data(grid3<Cell> &g) { return cell_count(g) ? &g[0] : nullptr; }
		template <typename Cell> const Cell *cell_data(const grid3<Cell> &g) { return cell_count(g) ? &g[0] : nullptr; }
		/// A mac_grid of the given size without the reference's out-of-line constructor (src/mac_grid.cpp:8-9), so that a host
		/// does not have to link that file: `_grid` is a protected member by the reference's design.
		struct sized_mac_grid : mac_grid {
			explicit sized_mac_grid(vec3s n) { _grid = grid3<cell>(n); }
		};
		inline mac_grid make_mac_grid(vec3s n) { return mac_grid(static_cast<const mac_grid &>(sized_mac_grid(n))); }
	}
}
