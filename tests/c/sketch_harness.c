/* Harness for the btbb_find_ac patch printed in INTEGRATION.md section 2 (test infrastructure).
 * The patch is written against the inside of the reference's bluetooth_packet.c, so the three
 * things it uses from there get local stand-ins under other names; the block itself is pasted
 * by tests/test_c_dropin.py into sketch_block.inc, unchanged. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <btbb.h>

typedef struct { uint32_t LAP; uint8_t ac_errors; } sketch_packet;
static sketch_packet *sketch_packet_new(void) { return (sketch_packet *)calloc(1, sizeof(sketch_packet)); }
static void sketch_init_packet(sketch_packet *p, uint32_t lap, uint8_t e) { p->LAP = lap; p->ac_errors = e; }

#define btbb_packet sketch_packet
#define btbb_packet_new sketch_packet_new
#define init_packet sketch_init_packet
#define btbb_find_ac sketch_find_ac
#include "sketch_block.inc"
#undef btbb_packet
#undef btbb_packet_new
#undef init_packet
#undef btbb_find_ac

int main(int argc, char **argv)
{
	FILE *f;
	long n;
	char *sym;
	int off = 0, window, r, bad = 0, found = 0;
	if (argc < 2 || !(f = fopen(argv[1], "rb")))
		return 2;
	fseek(f, 0, SEEK_END);
	n = ftell(f);
	fseek(f, 0, SEEK_SET);
	sym = (char *)malloc((size_t)n);
	if (fread(sym, 1, (size_t)n, f) != (size_t)n)
		return 2;
	fclose(f);
	if (btbb_init(2))
		return 3;
	window = (int)n - 63;
	/* the caller loop of the reference's users: first match, resume one past it */
	while (off < window) {
		sketch_packet *mine = NULL;
		btbb_packet *theirs = NULL;
		int a = sketch_find_ac(sym + off, window - off, LAP_ANY, 2, &mine);
		r = btbb_find_ac(sym + off, window - off, LAP_ANY, 2, &theirs);      /* the shipped drop-in */
		if (a != r || (r >= 0 && (mine->LAP != btbb_packet_get_lap(theirs) ||
					  mine->ac_errors != btbb_packet_get_ac_errors(theirs)))) {
			printf("MISMATCH at %d: %d vs %d\n", off, a, r);
			bad++;
		}
		if (theirs)
			btbb_packet_unref(theirs);
		free(mine);
		if (r < 0)
			break;
		printf("AC offset=%d\n", off + r);
		found++;
		off += r + 1;
	}
	printf("DONE found=%d bad=%d\n", found, bad);
	return bad ? 1 : 0;
}
