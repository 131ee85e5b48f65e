"""Re-flow a Markdown file to lines of at most WIDTH characters (default 118): paragraphs and list items (with a hanging indent)
are re-wrapped; headings, tables, fenced code blocks, indented code and blank lines are left as they are.
usage: python profiles/tools/rewrap_md.py FILE [WIDTH]     (rewrites FILE in place)"""
import re
import sys
import textwrap

path = sys.argv[1]
width = int(sys.argv[2]) if len(sys.argv) > 2 else 118
lines = open(path).read().split("\n")
out = []
para = []          # lines of the paragraph / list item being collected
first_prefix = rest_prefix = ""


def flush():
    global para
    if not para:
        return
    text = " ".join(s.strip() for s in para)
    w = textwrap.TextWrapper(width=width, initial_indent=first_prefix, subsequent_indent=rest_prefix, break_long_words=False,
                             break_on_hyphens=False)
    out.extend(w.wrap(text) or [first_prefix.rstrip()])
    para = []


in_code = False
item = re.compile(r"^(\s*)([-*]|\d+\.)\s+")
for ln in lines:
    if ln.strip().startswith("```"):
        flush(); in_code = not in_code; out.append(ln); continue
    if in_code:
        out.append(ln); continue
    if not ln.strip():
        flush(); out.append(""); continue
    if ln.lstrip().startswith(("#", "|")) or ln.startswith("    ") and not para:
        flush(); out.append(ln); continue
    m = item.match(ln)
    if m:
        flush()
        first_prefix = m.group(0)
        rest_prefix = " " * len(first_prefix)
        para = [ln[len(first_prefix):]]
        continue
    if not para:
        first_prefix = rest_prefix = re.match(r"^\s*", ln).group(0) if False else ""
        # a continuation paragraph inside a list keeps its indent
        ind = re.match(r"^\s*", ln).group(0)
        first_prefix = rest_prefix = ind
    para.append(ln)
flush()
open(path, "w").write("\n".join(out))
print(f"{path}: {len(lines)} -> {len(out)} lines, longest {max(len(s) for s in out)}")
