"""One-shot (round 5): split the 123 KB DESIGN.md into docs/ chapters with lines of at most 120 characters.
Paragraphs and list items are re-wrapped (hanging indent kept), fenced code is left alone, tables whose rows do not fit
are turned into record lists (one bullet per row, one sub-bullet per column).  usage: python tools/split_design.py OLD.md"""
import re
import sys
import textwrap

W = 120
CHAPTERS = [
    ("01_path_and_boundary.md", ["1."]),
    ("02_layout_in_hbm.md", ["2."]),
    ("03_kernels.md", ["3."]),
    ("04_oracle_and_parity.md", ["4."]),
    ("05_measurement.md", ["5."]),
    ("06_multi_gpu.md", ["6."]),
    ("07_widened_rows.md", ["7."]),
    ("08_scope_and_next.md", ["8.", "9."]),
    ("09_round_logs.md", ["10.", "11.", "12."]),
]


def wrap_block(lines):
    """one paragraph or list item (lines already joined logically)"""
    first = lines[0]
    m = re.match(r"^(\s*)([-*+]|\d+\.)\s+", first)
    text = " ".join(l.strip() for l in lines)
    if m:
        indent = m.group(1)
        bullet = m.group(2)
        body = text[len(bullet):].strip() if text.startswith(bullet) else text
        body = re.sub(r"^([-*+]|\d+\.)\s+", "", " ".join(l.strip() for l in lines))
        init = indent + bullet + " "
        sub = indent + " " * (len(bullet) + 1)
        return textwrap.wrap(body, width=W, initial_indent=init, subsequent_indent=sub, break_long_words=False,
                             break_on_hyphens=False) or [init.rstrip()]
    indent = re.match(r"^(\s*)", first).group(1)
    if first.lstrip().startswith(">"):
        return lines
    return textwrap.wrap(text, width=W, initial_indent=indent, subsequent_indent=indent, break_long_words=False,
                         break_on_hyphens=False) or [""]


def table_to_records(rows):
    cells = [[c.strip() for c in r.strip().strip("|").split("|")] for r in rows]
    # (escaped pipes inside cells are rare here; rows with a different cell count are kept verbatim)
    header = cells[0]
    body = [c for c in cells[2:]]
    out = []
    for c in body:
        if len(c) != len(header):
            out.append("| " + " | ".join(c) + " |")
            continue
        title = c[0] if c[0] else "(row)"
        lab0 = header[0] if header[0] else "item"
        out += textwrap.wrap(f"**{lab0}: {title}**", width=W, initial_indent="- ", subsequent_indent="  ", break_long_words=False,
                             break_on_hyphens=False)
        for h, v in zip(header[1:], c[1:]):
            if not v:
                continue
            out += textwrap.wrap(f"{h}: {v}" if h else v, width=W, initial_indent="  - ", subsequent_indent="    ",
                                 break_long_words=False, break_on_hyphens=False)
    return out


def rewrap(md):
    out, i, lines = [], 0, md.split("\n")
    while i < len(lines):
        l = lines[i]
        if l.startswith("```"):
            out.append(l)
            i += 1
            while i < len(lines) and not lines[i].startswith("```"):
                out.append(lines[i])
                i += 1
            if i < len(lines):
                out.append(lines[i])
                i += 1
            continue
        if l.startswith("#") or l.strip() == "":
            out.append(l)
            i += 1
            continue
        if l.lstrip().startswith("|"):
            j = i
            while j < len(lines) and lines[j].lstrip().startswith("|"):
                j += 1
            rows = lines[i:j]
            if max(len(r) for r in rows) <= W or len(rows) < 3:
                out += rows
            else:
                out += table_to_records(rows)
            i = j
            continue
        # paragraph / list item: collect until blank line, a new list item, heading, table or fence
        blk = [l]
        i += 1
        while i < len(lines):
            n = lines[i]
            if (n.strip() == "" or n.startswith("#") or n.startswith("```") or n.lstrip().startswith("|")
                    or re.match(r"^\s*([-*+]|\d+\.)\s+", n)):
                break
            blk.append(n)
            i += 1
        out += wrap_block(blk)
    return "\n".join(out)


def main():
    src = open(sys.argv[1]).read()
    parts = re.split(r"(?m)^(?=## )", src)
    head, secs = parts[0], parts[1:]
    for fname, keys in CHAPTERS:
        chosen = [s for s in secs if any(s.startswith("## " + k + " ") or s.startswith("## " + k) for k in keys)]
        body = "".join(chosen)
        title = chosen[0].split("\n", 1)[0][3:] if chosen else fname
        txt = f"<!-- chapter of DESIGN.md (round 5 split); lines <= {W} characters -->\n" + rewrap(body).rstrip() + "\n"
        open("docs/" + fname, "w").write(txt)
        longest = max((len(x) for x in txt.split("\n")), default=0)
        print(fname, len(txt), "bytes; longest line", longest, "|", title[:60])


if __name__ == "__main__":
    main()
