# oracle/ref_extract_functions.awk -- TEST INFRASTRUCTURE ONLY.  Prints, from a reference source file read where it lies, the top-level functions whose first line (or, for
# templates, the line after `template <...>`) contains one of the names in NAMES (separated by '|'), each from that line to its closing brace in column 1 (member functions `X::name(` are not taken).  Used by
# oracle/Makefile to feed selected kernels of the reference to g++ on a pipe; nothing is written to disk.
BEGIN { n = split(NAMES, want, "|") }
{
	if (!inside) {
		for (k = 1; k <= n; ++k) if (index($0, want[k] "(") > 0 && index($0, "::" want[k] "(") == 0 && $0 !~ /^[ \t]/ && $0 !~ /;[ \t]*$/) { inside = 1; if (prev ~ /^template/) print prev; break }
	}
	if (inside) { print; if ($0 ~ /^}/) inside = 0 }
	prev = $0
}
